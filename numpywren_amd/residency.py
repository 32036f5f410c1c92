"""HBM residency of stored tiles: which tiles of the object table stay in device memory and which live in pinned
host DRAM.

The reference keeps every tile in S3 and caches a handful per worker (LRUCache, reference job_runner.py:34-64); here
the store itself is HBM (288 GB per MI355X) and host DRAM is the overflow tier.  Policy: while a program runs, the
executor installs a `SpillPlan` -- the static task sequence's list of who reads which stored tile when -- and the victim
is the tile whose next read is FARTHEST in that sequence (Belady's rule; a tile no remaining task reads goes first), the
spilled tiles the next few tasks read are copied back ahead of them (`prefetch`), and only without a plan does the
least-recently-used tile go first.  The reference overlaps tile reads with compute by construction (a `read` stage
`pipeline_width` tasks ahead of `compute`, reference job_runner.py:224-254) and caches LRU (:34-64); with the DAG known in
advance both the victim and the moment of the copy-in are known, too.  A Cholesky sweeps its trailing matrix once per step
-- the access pattern on which LRU misses every tile: 32768^2 with 24 tiles of budget moves 106 tiles out and 144 back
under LRU, 56 and 19 under the plan.  Two triggers:

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


class SpillPlan(object):
    """Who reads which stored tile when, from the expanded DAG: `order` is the predicted issue order of the tasks (the ready
    heap's: longest path to a sink first), `key_of(matrix name, tile index)` the table key of a tile.  The executor reports
    every task it issues (`issued`); the next read of a tile is then the first position in its list that has not been
    issued -- exact whatever the real order (batches, the chain partition) does to the predicted one."""

    def __init__(self, order, key_of):
        self.pending = {}       # table key -> positions (ascending) of the not yet issued tasks that read it
        self.reads_at = []      # position -> table keys that task reads
        self.pos = {}           # task index -> position
        self.done = []          # position -> issued?
        self.cursor = 0         # lowest position not issued yet
        for p, t in enumerate(order):
            self.pos[t.index] = p
            keys = [key_of(*r) for r in dict.fromkeys(t.reads)]
            self.reads_at.append(keys)
            self.done.append(False)
            for k in keys:
                self.pending.setdefault(k, []).append(p)

    NEVER = 1 << 60

    def next_use(self, tkey):
        lst = self.pending.get(tkey)
        return lst[0] if lst else self.NEVER

    def issued(self, task_index):
        p = self.pos.get(task_index)
        if p is None or self.done[p]:
            return
        self.done[p] = True
        for k in self.reads_at[p]:
            lst = self.pending.get(k)
            if lst:
                try:
                    lst.remove(p)
                except ValueError:
                    pass
        while self.cursor < len(self.done) and self.done[self.cursor]:
            self.cursor += 1

    def upcoming(self, window):
        """Table keys the next `window` not yet issued tasks read, nearest first, each once."""
        seen, out, p, left = set(), [], self.cursor, window
        while p < len(self.done) and left > 0:
            if not self.done[p]:
                left -= 1
                for k in self.reads_at[p]:
                    if k not in seen:
                        seen.add(k)
                        out.append(k)
            p += 1
        return out


class Residency(object):
    """Bookkeeping over the DeviceTiles of one object table: plan-driven (SpillPlan) while a program runs, least recently
    used otherwise.  All methods run under the table's lock."""

    def __init__(self, table):
        self.table = table
        self.lru = collections.OrderedDict()  # (bucket, key_base, tile_key) -> (id of the tile's DeviceBuffer, offset)
        self.bufs = {}                         # (id(DeviceBuffer), offset) -> [nbytes, {table keys}]
        self.resident_bytes = 0
        self._budget = None
        self._budget_known = False
        self.evictions = 0
        self.restores = 0
        self.prefetched = 0
        self.written_through = 0
        self._hooked = None
        self.plan = None          # SpillPlan of the program being run (LambdaPackExecutor installs it)
        self.plan_factory = None  # ... or how to make it, when the tier was idle as the run began (called at the first need)
        self.last_policy = "lru"  # what chose the victims of the most recent run ("plan" once a plan was installed; the plan
                                  # itself is dropped when its run ends -- job_runner.release_spill_plan)

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
            self.evictions = self.restores = self.prefetched = self.written_through = 0
            self.plan = self.plan_factory = None
            self.last_policy = "lru"

    def stats(self):
        return {"resident_bytes": self.resident_bytes, "resident_tiles": len(self.lru), "budget": self._budget,
                "evictions": self.evictions, "restores": self.restores, "prefetched": self.prefetched, "written_through": self.written_through,
                "policy": "plan" if self.plan is not None else self.last_policy}

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

    def write_through(self, tkey, obj, be):
        """A tile that has just been stored and cannot stay until its next read -- the resident tiles that are read SOONER
        already fill the budget, so the plan's rule will push it out first -- starts its copy to pinned host memory NOW,
        behind its producer, and keeps it as its host copy: pushing it out later is then free.  Why not wait for the
        eviction: victims leave farthest-next-read first, i.e. in the REVERSE of the order the next step reads them back,
        and on one in-order copy stream the first tile wanted back would sit behind every other tile's copy out (measured,
        32768^2 with 12 tiles of budget: the copies in and out took turns instead of overlapping, tools/spill_timeline.py).
        In production order the copies out run in the order the copies back will be asked for."""
        plan, budget = self.plan, self.budget
        if plan is None or budget is None or not self._evictable(obj):
            return False
        if not hasattr(obj.buf, "aux") or not hasattr(be, "spill_to_host"):
            return False        # (a backend whose buffers carry no host copies: pushing out later copies)
        aux = obj.buf.aux
        offset = getattr(obj, "offset", 0)   # (a batched kernel's outputs share one allocation: one host copy per tile of it)
        if isinstance(aux, dict) and (aux.get("host_copies") or {}).get(offset) is not None:
            return False
        mine = plan.next_use(tkey)
        sooner = obj.nbytes
        for bid, ent in self.bufs.items():
            if tkey in ent[1]:
                continue
            if min(plan.next_use(k) for k in ent[1]) < mine:
                sooner += ent[0]
                if sooner > budget:
                    break
        if sooner <= budget:
            return False
        sp = be.spill_to_host(obj)
        aux = dict(aux or {})
        aux["host_copies"] = dict(aux.get("host_copies") or {})
        aux["host_copies"][offset] = sp
        obj.buf.aux = aux
        self.written_through += 1
        return True

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

    def _victims(self):
        """Resident table keys, first victim first: with a plan the tile whose next read is farthest in the task sequence
        (never read again: first; ties in least-recently-used order), without one the least recently used."""
        keys = list(self.lru.keys())
        if self.plan is None and self.plan_factory is not None:
            factory, self.plan_factory = self.plan_factory, None
            try:
                factory()
            except Exception:
                pass
        plan = self.plan
        if plan is None:
            return keys
        # a buffer stored under several keys is needed as soon as any of them is read
        def soonest(k):
            ent = self.bufs.get(self.lru[k])
            return min(plan.next_use(x) for x in (ent[1] if ent else (k,)))
        return sorted(keys, key=lambda k: -soonest(k))     # (stable: equal distances keep the LRU order)

    def enforce(self, protect=()):
        """Evict tiles (see _victims) until the stored tiles fit the budget.  `protect`: keys to keep."""
        budget = self.budget
        if budget is None or self.resident_bytes <= budget:
            return 0
        from .device import get_backend
        be = get_backend()
        freed = 0
        for tkey in self._victims():
            if self.resident_bytes <= budget:
                break
            freed += self._evict(tkey, be, protect)
        return freed

    def prefetch(self, be, window):
        """Copy the spilled tiles the next `window` tasks of the plan read back into HBM now, on the inbound spill stream,
        nearest reader first -- ahead of the tasks instead of inside their get_tile.  Under a budget a tile only comes
        back early if what it pushes out is needed later than itself."""
        plan = self.plan
        if plan is None or window <= 0:
            return 0
        from .device import SpilledTile
        n = 0
        with self.table.lock:
            for tkey in plan.upcoming(window):
                obj = self._lookup(tkey)
                if not isinstance(obj, SpilledTile):
                    continue
                budget = self.budget
                if budget is not None and self.resident_bytes + obj.nbytes > budget:
                    mine = plan.next_use(tkey)
                    victims = [k for k in self._victims()[:1] if k != tkey]
                    if not victims or plan.next_use(victims[0]) <= mine:
                        break
                self.restore(tkey, obj, be)
                self.prefetched += 1
                n += 1
        return n

    def reclaim(self, nbytes):
        """Out-of-memory handler of the allocator: push out at least `nbytes` of least-recently-used tiles."""
        from .device import get_backend
        be = get_backend()
        freed = 0
        with self.table.lock:
            for tkey in self._victims():
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
