"""tl2.proj.logger.logging_utils_v2.get_logger (scripts/dataset_tool.py:26, 367: a plain file logger for an image list)"""
import logging
import os


def get_logger(filename=None, logger_names=(), stream=True, mode='w', level=logging.INFO, **kwargs):
    name = f"tl2shim.{filename}"
    logger = logging.getLogger(name)
    logger.setLevel(level)
    logger.propagate = False
    logger.handlers.clear()
    if filename:
        os.makedirs(os.path.dirname(os.path.abspath(filename)), exist_ok=True)
        h = logging.FileHandler(filename, mode=mode)
        h.setFormatter(logging.Formatter("%(message)s"))
        logger.addHandler(h)
    if stream:
        logger.addHandler(logging.StreamHandler())
    return logger
