"""Processor interface (mirror of src/dsp/data_processor.py:23-41)."""
from abc import ABC, abstractmethod


class DataProcessor(ABC):
    """A consumer that takes chunks from a queue until ``isDead`` is set and writes results to ``f``."""

    @abstractmethod
    def processData(self, *args, **kwargs) -> None:
        """processData(isDead, buffer, f): ``isDead`` a multiprocessing.Value('b'), ``buffer`` a Queue of chunks,
        ``f`` an output file name or None (stdout)."""
        pass
