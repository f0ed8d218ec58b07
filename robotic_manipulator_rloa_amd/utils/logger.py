"""Framework logger: one named logger with the reference's level API (utils/logger.py:62-68 get_global_logger;
:11-27 a stdout handler + a rotating `training_logs.log`). The file handler is opt-in here
(NAF_LOG_FILE=1 or Logger.set_logger_setup(log_file=True)): importing the package must not create files in CWD."""
import logging
import os
import sys
from logging.handlers import RotatingFileHandler

_NAME = 'robotic_manipulator_rloa.utils.logger'   # same logger name as the reference, so user filters keep working


def get_global_logger() -> logging.Logger:
    return logging.getLogger(_NAME)


class Logger:
    _configured = False

    @staticmethod
    def set_logger_setup(log_file: bool = None) -> None:
        logger = get_global_logger()
        if Logger._configured:
            return
        Logger._configured = True
        h = logging.StreamHandler(sys.stdout)
        h.setFormatter(logging.Formatter('%(asctime)s | %(levelname)-8s | %(message)s'))
        logger.addHandler(h)
        logger.propagate = False
        if log_file is None:
            log_file = os.environ.get('NAF_LOG_FILE') == '1'
        if log_file:
            fh = RotatingFileHandler(filename='training_logs.log', mode='a', maxBytes=50000000, backupCount=10,
                                     encoding='utf-8')
            fh.setFormatter(logging.Formatter('"%(levelname)s"|"%(asctime)s"|%(message)s'))
            fh.setLevel(logging.INFO)
            logger.addHandler(fh)
        logger.setLevel(20)
