"""HBM residency of stored tiles: which tiles of the object table stay in device memory and which live in pinned
host DRAM.

The reference keeps every tile in S3 and caches a handful per worker (LRUCache, reference job_runner.py:34-64); here
the store itself is HBM (288 GB per MI355X) and host DRAM is the overflow tier.  Policy: least-recently-used stored
tile goes first.  Two triggers:

  * a byte budget for stored tiles (`store.hbm_budget_bytes` / $NUMPYWREN_AMD_HBM_BUDGET, e.g. "200G"); unset = no
    budget, and
  * an allocation failure on the device: the backend calls `reclaim(nbytes)` (registered as an out-of-memory handler)
    before it gives up.

Mechanics: eviction starts an asynchronous D2H copy into pinned memory on the backend's spill stream (ordered after
the tile's producer) and swaps the table entry for the `SpilledTile`; tasks that already hold the DeviceTile keep
using it, the HBM goes back to the pool when the last of them lets go.  A later `get_tile` copies the bytes back on
the same stream (so it is ordered behind the D2H) into a fresh DeviceTile whose `ready` event consumers wait on, and
the tile is resident again.  Stored tiles are immutable (a `put` replaces the object), so the bytes that come back are
the bytes that left.
"""
import collections
import os
import re

_SUFFIX = {"k": 1 << 10, "m": 1 << 20, "g": 1 << 30, "t": 1 << 40}


def parse_bytes(value):
    """None | int | "200G" / "512M" / "1.5T" / "1234" -> bytes (or None)."""
    if value is None:
        return None
    if isinstance(value, (int, float)):
        return int(value)
    s = str(value).strip().lower()
    if s in ("", "none"):
        return None
    m = re.fullmatch(r"([0-9]*\.?[0-9]+)\s*([kmgt]?)(i?b)?", s)
    if m is None:
        raise ValueError("not a byte count: {0!r} (expected e.g. 200G, 512M, 1.5T or a plain number)".format(value))
    return int(float(m.group(1)) * _SUFFIX.get(m.group(2), 1))


class Residency(object):
    """LRU bookkeeping over the DeviceTiles of one object table.  All methods run under the table's lock."""

    def __init__(self, table):
        self.table = table
        self.lru = collections.OrderedDict()  # (bucket, key_base, tile_key) -> (id of the tile's DeviceBuffer, offset)
        self.bufs = {}                         # (id(DeviceBuffer), offset) -> [nbytes, {table keys}]
        self.resident_bytes = 0
        self._budget = None
        self._budget_known = False
        self.evictions = 0
        self.restores = 0
        self._hooked = None

    # ---- configuration ----
    @property
    def budget(self):
        if not self._budget_known:
            from . import config
            b = os.environ.get("NUMPYWREN_AMD_HBM_BUDGET")
            if b is None:
                b = config.default()["store"].get("hbm_budget_bytes")
            self._budget = parse_bytes(b)
            self._budget_known = True
        return self._budget

    def set_budget(self, nbytes):
        """Byte budget for stored tiles (None: unlimited); applied immediately."""
        with self.table.lock:
            self._budget = parse_bytes(nbytes)
            self._budget_known = True
            self.enforce()

    def reset(self):
        with self.table.lock:
            self.lru.clear()
            self.bufs.clear()
            self.resident_bytes = 0
            self._budget_known = False
            self.evictions = self.restores = 0

    def stats(self):
        return {"resident_bytes": self.resident_bytes, "resident_tiles": len(self.lru), "budget": self._budget,
                "evictions": self.evictions, "restores": self.restores}

    def _hook(self, be):
        # the allocator asks us for memory before it reports out-of-memory
        if self._hooked is not be and hasattr(be, "oom_handlers"):
            be.oom_handlers.append(self.reclaim)
            self._hooked = be

    # ---- bookkeeping ----
    def _evictable(self, obj):
        from .device import DeviceTile
        return isinstance(obj, DeviceTile) and not obj.shared and obj.nbytes > 0

    def note_put(self, tkey, obj):
        """`obj` has just been stored under `tkey` (replacing whatever was there)."""
        self.note_delete(tkey)
        if not self._evictable(obj):
            return
        bid = (id(obj.buf), getattr(obj, "offset", 0))
        ent = self.bufs.get(bid)
        if ent is None:
            ent = self.bufs[bid] = [obj.nbytes, set()]
            self.resident_bytes += obj.nbytes
        ent[1].add(tkey)
        self.lru[tkey] = bid

    def note_delete(self, tkey):
        bid = self.lru.pop(tkey, None)
        if bid is None:
            return
        ent = self.bufs.get(bid)
        if ent is not None:
            ent[1].discard(tkey)
            if not ent[1]:
                self.resident_bytes -= ent[0]
                del self.bufs[bid]

    def touch(self, tkey):
        if tkey in self.lru:
            self.lru.move_to_end(tkey)

    # ---- policy ----
    def _lookup(self, tkey):
        d = self.table.objects.get((tkey[0], tkey[1]))
        return None if d is None else d.get(tkey[2])

    def _evict(self, tkey, be, protect=()):
        """Move the buffer stored under `tkey` (and under every other key that shares it) to pinned host memory.
        Returns the HBM bytes that become reclaimable."""
        bid = self.lru.get(tkey)
        ent = self.bufs.get(bid)
        if ent is None:
            self.lru.pop(tkey, None)
            return 0
        keys = list(ent[1])
        if any(k in protect for k in keys):
            return 0
        live = [(k, self._lookup(k)) for k in keys]
        live = [(k, o) for k, o in live if self._evictable(o) and (id(o.buf), getattr(o, "offset", 0)) == bid]
        for k in keys:
            self.note_delete(k)   # also drops entries the table no longer holds (cleared behind our back)
        if not live:
            return 0
        spilled = {}
        for k, o in live:
            # one D2H per buffer; reshaped handles on the same buffer get their own shape on the same host bytes
            sp = spilled.get(o.shape)
            if sp is None:
                first = next(iter(spilled.values()), None)
                if first is None:
                    sp = be.spill_to_host(o)
                else:
                    sp = type(first)(first.buf, o.shape, o.dtype, first.ready)
                spilled[o.shape] = sp
            self.table.objects[(k[0], k[1])][k[2]] = sp
        self.evictions += 1
        return live[0][1].nbytes

    def enforce(self, protect=()):
        """Evict least-recently-used tiles until the stored tiles fit the budget.  `protect`: keys to keep."""
        budget = self.budget
        if budget is None or self.resident_bytes <= budget:
            return 0
        from .device import get_backend
        be = get_backend()
        freed = 0
        for tkey in list(self.lru.keys()):
            if self.resident_bytes <= budget:
                break
            freed += self._evict(tkey, be, protect)
        return freed

    def reclaim(self, nbytes):
        """Out-of-memory handler of the allocator: push out at least `nbytes` of least-recently-used tiles."""
        from .device import get_backend
        be = get_backend()
        freed = 0
        with self.table.lock:
            for tkey in list(self.lru.keys()):
                if freed >= nbytes:
                    break
                freed += self._evict(tkey, be)
        return freed

    def restore(self, tkey, spilled, be):
        """Bring a spilled tile back into HBM, make it the table entry again and return the DeviceTile."""
        self._hook(be)
        tile = be.restore_from_host(spilled)
        d = self.table.objects.get((tkey[0], tkey[1]))
        if d is not None and d.get(tkey[2]) is spilled:
            d[tkey[2]] = tile
            self.note_put(tkey, tile)
            self.restores += 1
            self.enforce(protect=(tkey,))
        return tile
