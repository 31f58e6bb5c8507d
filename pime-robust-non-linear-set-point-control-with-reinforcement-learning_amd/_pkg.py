"""Top-level names of the pime_amd package."""
__version__ = "0.1.0"

PACKAGE_DIR = __import__("os").path.dirname(__import__("os").path.abspath(__file__))
