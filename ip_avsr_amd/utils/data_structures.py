"""Fixed-capacity FIFO window (reference utils/data_structures.py:1-48)."""


class circular_list(object):
    def __init__(self, size, init=None):
        self.MAX_SIZE = size
        self._data = [init] * size if init is not None else []

    def push(self, item):
        """Append at the tail, evicting the head when full."""
        if len(self._data) == self.MAX_SIZE:
            del self._data[0]
        self._data.append(item)

    def pop(self):
        """Remove and return the head (None when empty)."""
        return self._data.pop(0) if self._data else None

    def __iter__(self):
        return iter(list(self._data))

    def __getitem__(self, index):
        return self._data[index]

    def __setitem__(self, index, value):
        self._data[index] = value

    def __len__(self):
        return len(self._data)
