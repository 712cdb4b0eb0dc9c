"""Mirror of reference util/timing.py:6-12."""
import contextlib
import logging
import time


@contextlib.contextmanager
def log_duration(operation):
    logging.info(operation)
    start_time = time.time()
    yield
    logging.debug(f"{operation} took {time.time() - start_time:.2f} seconds")
