"""Reader for the record databases of the reference's LMDB-cached datasets (second half).

The reference writes a dataset ONCE into an LMDB environment -- for the paper's "N x CIFAR" runs `rounds` augmented passes over the base
dataset, one after the other -- and then trains from that static database (``fullbatch/data/lmdb_datasets.py``: writer ``_create_database``
:198-300, reader ``LMDBDataset`` :28-162).  Record layout (what this module reads):

    b"0", b"1", ...      one image each: the raw bytes of a uint8 array of shape ``__shape__`` (CHW when the dataset's first live
                         transform was ToTensor, else HWC), in writing order; ``rounds * len(dataset)`` entries
    b"__keys__"          pickle of the list of keys          b"__labels__"   pickle of the list of int labels (one per entry)
    b"__len__"           pickle of the entry count           b"__shape__"    pickle of the per-image shape

A static, written-once database is exactly what the full-batch engine wants: ``LMDBRecords.as_feed()`` decodes it into the
``(images, labels)`` tensor pair that ``train()`` takes, with the reference's per-item arithmetic (``uint8 / 255`` then the live
``Normalize``), and every rank copies only its own chunk range to the GPU.

The key-value store itself is the third-party ``lmdb`` package (pinned by the reference as a dependency; absent from the build image
-- ``open()`` then fails loudly).  Any object with its read API (``begin()`` -> transaction with ``get`` and ``cursor``) can be passed as
``env``; the tests pass the store the REFERENCE's own writer filled (``tests/golden/make_golden.py --r2``).
"""
import pickle

import numpy as np
import torch


class LMDBRecords:
    def __init__(self, path=None, env=None, access="get", max_readers=128, readahead=False, meminit=False, max_spare_txns=128):
        """``path``: the database file (the reference opens it with ``subdir=False``, read-only, no lock -- reference :60-69);
        ``env``: an already opened environment instead.  ``access``: "get" or "cursor" (reference ``cfg_db.access``, :139-148)."""
        if env is None:
            if path is None:
                raise ValueError("LMDBRecords needs a database path or an opened environment")
            try:
                import lmdb
            except ImportError as exc:
                raise RuntimeError("reading an LMDB database file needs the `lmdb` package (a dependency of the reference, setup.cfg:32-37); "
                                   "it is not installed here") from exc
            env = lmdb.open(path, subdir=False, max_readers=max_readers, readonly=True, lock=False, readahead=readahead, meminit=meminit,
                            max_spare_txns=max_spare_txns)
        self.env, self.access, self.path = env, access, path
        try:
            with self.env.begin(write=False) as txn:
                self.length = pickle.loads(txn.get(b"__len__"))
                self.keys = pickle.loads(txn.get(b"__keys__"))
                self.labels = pickle.loads(txn.get(b"__labels__"))
                self.shape = tuple(pickle.loads(txn.get(b"__shape__")))
        except TypeError as exc:          # txn.get() returned None: the writer has not finished (reference :76-80 waits and retries)
            raise ValueError(f"The LMDB dataset at {path} is unfinished or damaged (no __len__/__keys__/__labels__/__shape__ records).") from exc
        if len(self.keys) != self.length or len(self.labels) != self.length:
            raise ValueError(f"LMDB dataset at {path}: {self.length} entries, {len(self.keys)} keys, {len(self.labels)} labels")
        self._txn = self._cursor = None

    def __len__(self):
        return self.length

    @property
    def channels_first(self):
        """CHW records (written when the dataset's first live transform was ToTensor) or HWC (reference :40-51, :277-280)."""
        return self.shape[0] in (1, 3) and self.shape[-1] not in (1, 3)

    def _bytes(self, index):
        if self.access == "cursor":                              # reference :139-146
            if self._cursor is None:
                self._txn = self.env.begin(write=False)
                self._cursor = self._txn.cursor()
                self._cursor.first()
            key = "{}".format(index).encode("ascii")
            if key != self._cursor.key():
                self._cursor.set_key(key)
            data = self._cursor.value()
            self._cursor.next()
            return data
        with self.env.begin(write=False) as txn:                 # reference :147-148
            return txn.get(self.keys[index])

    def __getitem__(self, index):
        """(uint8 tensor of shape ``self.shape``, int label): the reference's ``data_block`` / ``label`` before any transform (:150-160)."""
        block = torch.frombuffer(bytearray(self._bytes(index)), dtype=torch.uint8).view(self.shape)
        return block, self.labels[index]

    def images_uint8(self):
        """All records as one uint8 tensor [N, C, H, W] in database order."""
        out = np.empty((self.length,) + self.shape, dtype=np.uint8)
        flat = out.reshape(self.length, -1)
        for i in range(self.length):
            flat[i] = np.frombuffer(self._bytes(i), dtype=np.uint8)
        x = torch.from_numpy(out)
        return x if self.channels_first else x.permute(0, 3, 1, 2).contiguous()

    def as_feed(self, mean=None, std=None):
        """The ``(images, labels)`` pair for ``train()``: float images ``uint8 / 255`` (ToTensor semantics, reference :153-156) and, with
        ``mean`` / ``std``, the live ``Normalize`` the reference applies per item; labels int64.  N x CIFAR databases simply have
        ``rounds`` times the entries of the base dataset."""
        x = self.images_uint8().to(torch.float) / 255
        if mean is not None:
            m = torch.as_tensor(mean, dtype=torch.float).view(1, -1, 1, 1)
            s = torch.as_tensor(std, dtype=torch.float).view(1, -1, 1, 1)
            x = (x - m) / s
        return x, torch.as_tensor(self.labels, dtype=torch.long)
