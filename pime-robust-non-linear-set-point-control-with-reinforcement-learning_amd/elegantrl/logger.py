"""Metrics sink with the call surface the reference expects from its (missing) `elegantrl/logger.py`:
record / dump / configure / Figure / make_output_format.  Keys used by the run loop:
rollout/ep_rew_mean, rollout/ep_rew_std, rollout/log_rew_max, train/*_loss, train/n_updates,
training/total_step, plus perf/env_steps_per_s added by this build.  Output: JSON lines and a CSV under `folder`."""
import csv
import json
import os
import time

_state = {"folder": None, "values": {}, "csv_keys": None, "t0": time.time()}


class Figure:
    def __init__(self, figure=None, close=True):
        self.figure, self.close = figure, close


def configure(folder=None, format_strings=None):
    _state["folder"] = folder
    _state["values"] = {}
    _state["csv_keys"] = None
    if folder:
        os.makedirs(folder, exist_ok=True)
    return _state


def record(key, value, exclude=None):
    _state["values"][key] = value


def get(key, default=None):
    return _state["values"].get(key, default)


def dump(step=0):
    vals = dict(_state["values"])
    vals["step"] = step
    vals["time/elapsed_s"] = round(time.time() - _state["t0"], 3)
    folder = _state["folder"]
    if folder:
        flat = {k: (float(v) if isinstance(v, (int, float)) or hasattr(v, "__float__") else str(v)) for k, v in vals.items()}
        with open(os.path.join(folder, "progress.jsonl"), "a") as f:
            f.write(json.dumps(flat) + "\n")
        path = os.path.join(folder, "progress.csv")
        keys = sorted(flat)
        new = _state["csv_keys"] != keys
        with open(path, "a", newline="") as f:
            w = csv.DictWriter(f, fieldnames=keys)
            if new:
                w.writeheader()
                _state["csv_keys"] = keys
            w.writerow(flat)
    _state["values"] = {}
    return vals


def make_output_format(_format, log_dir, log_suffix=""):
    return None
