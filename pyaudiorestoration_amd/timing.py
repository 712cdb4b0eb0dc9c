"""`log_duration` with the call contract of the reference's util/timing.py:6-12 (INFO line on entry, DEBUG line
with the elapsed seconds on exit) -- used by resampling.run exactly like the reference uses its own."""
import logging
from time import perf_counter


class log_duration:
    """with log_duration("Resampling"): ...   (re-usable, exception-transparent)"""

    def __init__(self, operation):
        self.operation = operation
        self.elapsed = None

    def __enter__(self):
        logging.info(self.operation)
        self._t0 = perf_counter()
        return self

    def __exit__(self, exc_type, exc, tb):
        self.elapsed = perf_counter() - self._t0
        logging.debug("%s took %.2f seconds", self.operation, self.elapsed)
        return False
