# The importable name of this directory is `pime_amd` (see /pime_amd/__init__.py at the repo root).
