"""configure_logger (interface of /root/reference/elegantrl/utils.py:10-46) on top of the local metrics sink."""
import glob
import os

from . import logger


def get_latest_run_id(log_path="", log_name=""):
    ids = [0]
    for path in glob.glob(os.path.join(log_path, f"{glob.escape(log_name)}_[0-9]*")):
        tail = path.split(os.sep)[-1].split("_")[-1]
        if tail.isdigit():
            ids.append(int(tail))
    return max(ids)


def configure_logger(verbose=0, tensorboard_log=None, tb_log_name="", reset_num_timesteps=True):
    if tensorboard_log is None:
        logger.configure(None)
        return
    latest = get_latest_run_id(tensorboard_log, tb_log_name)
    if not reset_num_timesteps:
        latest -= 1
    logger.configure(os.path.join(tensorboard_log, f"{tb_log_name}_{latest + 1}"), ["csv", "json"])
