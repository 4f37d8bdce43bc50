"""Item selectors of the per-user fine-tuning drivers (reference: cats.py, used by adaptive_test.py:76,87).

A selector is built from the items a user may be asked about and hands them out one at a time through
``next_item()``; ``asked`` records the order.  Kept: ``Next`` (in the order given - the one adaptive_test.py uses),
``Random`` and ``Popular``.  ``Fisher`` (cats.py:39-49) scores items from the ordinal model's cdf / pdf, which the live graph
never builds (the ordinal branch is dead code), so it is not offered here.
"""
import random as _random

import numpy as np


class CAT(object):
    def __init__(self, items, popularity=None):
        self.available_item_ids = [int(i) for i in items]
        self.asked = []
        self.popularity = None if popularity is None else np.asarray(popularity)

    def __len__(self):
        return len(self.available_item_ids)

    def _take(self, pos):
        item = self.available_item_ids.pop(pos)
        self.asked.append(item)
        return item

    def next_item(self):
        raise NotImplementedError


class Next(CAT):
    """cats.py:52-54: the first item not asked yet."""
    def next_item(self):
        return self._take(0)


class Random(CAT):
    """cats.py:23-28: a uniformly random remaining item (Python's global `random`, as the reference)."""
    def next_item(self):
        return self._take(_random.randrange(len(self.available_item_ids)))


class Popular(CAT):
    """cats.py:31-35: the remaining item with the highest popularity count."""
    def __init__(self, items, popularity):
        super(Popular, self).__init__(items, popularity)
        if self.popularity is None:
            raise ValueError("Popular needs a popularity array indexed by item id")

    def next_item(self):
        return self._take(int(self.popularity[self.available_item_ids].argmax()))


class Fisher(CAT):
    def __init__(self, *a, **kw):
        raise NotImplementedError("Fisher selection needs the ordinal model's cdf / pdf (cats.py:13-20,39-49); the live graph "
                                  "builds the binary / regression head only")
