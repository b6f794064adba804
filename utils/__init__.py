"""Alias package for the hot-path part of the reference's `utils` (utils_walkgpt.py, matcher.py); datasets, conversation templates and
logging helpers of the reference are out of scope and stay the reference's own files."""
