"""Input staging for the per-slide loops: stored fp16 bag -> async H2D on a copy stream, overlapped with the compute of
the previous bags -> the fused kernels, which convert fp16 -> fp32 / split-f16 in registers.

Reference behaviour replaced (own code): `HDF5_feat_dataset2` hands out fp16 bags (Step2_feature_extract.py:165 stores
fp16) and the loop does `data['input'].to(device, dtype=torch.float32)` per slide
(Step3_WSI_classification_ACMIL.py:193, :254): the fp16 -> fp32 widening runs on ONE host core (48.5 ms for a
50 000 x 512 bag, measured on the MI355X box), then 2x the PCIe bytes are copied synchronously.  Here:
  * the bag crosses PCIe in its stored dtype (fp16: 51 MB instead of 102 MB at N=50 000, D=512), nothing is converted
    on the host;
  * a background thread reads item i+1.. and issues its H2D on a dedicated HIP copy stream into a small ring of device
    buffers (grown to the largest bag seen, then reused: no per-slide allocation) while the main thread enqueues the
    kernels of item i.  The copy is taken straight from the tensor the dataset returned: the runtime's pageable-memory
    path sustains 46.8 GB/s here, whereas an extra host memcpy into a pinned ring ran at 3.7 GB/s and was dropped
    (a dataset that returns pinned tensors gets a fully asynchronous copy for free);
  * ordering is by events only: the compute stream waits on the slot's "copied" event, the copy stream waits on the
    slot's "consumed" event before the slot is overwritten; the host never synchronises with the GPU in steady state and
    runs at most `depth` bags ahead.
On a CPU device (the gloo tests) the prefetcher degrades to plain iteration with the same interface.
"""
from __future__ import annotations

import queue
import threading
from typing import Dict, Iterable, Iterator, Optional, Sequence

import torch


class _Slot:
    def __init__(self):
        self.dev: Optional[torch.Tensor] = None      # device, flat, dtype of the stored bags
        self.copied = None                           # torch.cuda.Event: H2D finished
        self.consumed = None                         # torch.cuda.Event: the kernels that read `dev` have finished


class BagPrefetcher:
    """Iterate `dataset[i] for i in order` yielding {'input': device tensor [N,D] (stored dtype), 'label': int, 'index': i}.

    depth = device slots (>= 2: one being computed, one being filled).  The yielded 'input' aliases a ring slot: it is
    valid until the NEXT item is requested (its slot is then handed back to the copy thread), which is what a
    one-slide-per-step loop needs; `.clone()` it to keep it longer.
    """

    def __init__(self, dataset, order: Sequence[int], device: torch.device, depth: int = 3, max_bytes: Optional[int] = None):
        self.dataset, self.order, self.device = dataset, list(order), torch.device(device)
        self.depth = max(2, depth)
        self.max_bytes = max_bytes            # budget for the device memory of ALL ring slots (None: `depth` slots whatever their size)
        self.alloc_bytes = 0                  # device bytes the ring holds now
        self.peak_bytes = 0                   # ... and the most it ever held
        self.slots_made = 0
        self.cuda = self.device.type == "cuda"
        self._ready: "queue.Queue" = queue.Queue()
        self._free: "queue.Queue" = queue.Queue()
        self._stop = False
        if self.cuda:
            self.copy_stream = torch.cuda.Stream(device=self.device)      # slots are created by the reader as bags need them (_take_slot)
        else:
            for _ in range(self.depth):
                self._free.put(None)
        self._thread = threading.Thread(target=self._reader, daemon=True)
        self._thread.start()

    def __len__(self):
        return len(self.order)

    def close(self):
        self._stop = True
        self._free.put(None)

    def _take_slot(self, need: int):
        """A ring slot for a bag of `need` bytes (copy thread).  Free slots are reused first; a new slot is made while fewer than
        `depth` exist and the ring stays inside `max_bytes` (the first two are unconditional: one computing, one filling); otherwise
        wait for a slot to come back -- after telling the consumer (a "flush" marker) so that a group it is still filling is launched
        with what it holds instead of waiting for bags that cannot be staged."""
        try:
            return self._free.get_nowait()
        except queue.Empty:
            pass
        grown = int(need * 1.25)
        if self.slots_made < self.depth and (self.slots_made < 2 or self.max_bytes is None or self.alloc_bytes + grown <= self.max_bytes):
            self.slots_made += 1
            return _Slot()
        self._ready.put(("flush",))
        return self._free.get()

    # ---- copy thread: read the item, wait (on the stream) for a free slot, issue the H2D on the copy stream
    def _reader(self):
        try:
            if self.cuda:
                torch.cuda.set_device(self.device)
            for i in self.order:
                item = self.dataset[i]                       # disk / decompression latency off the critical path
                x = item["input"]
                if not self.cuda:
                    self._free.get()                         # back-pressure: at most `depth` bags in flight
                    if self._stop:
                        break
                    self._ready.put((None, x, int(item["label"]), i))
                    continue
                n = x.numel()
                need = n * x.element_size()
                grown_b = int(n * 1.25) * x.element_size()
                while True:
                    slot = self._take_slot(need)              # back-pressure: at most `depth` bags / `max_bytes` of ring in flight
                    if self._stop or slot is None:
                        break
                    # The budget holds on REGROWTH too (slots created for small bags were later regrown to the largest bag
                    # without a look at max_bytes).  A recycled slot whose regrowth would leave the budget is retired -- its buffer goes
                    # back to the allocator once its readers are done -- and the ring continues with one slot fewer (never fewer than two).
                    old_b = 0 if slot.dev is None else slot.dev.numel() * slot.dev.element_size()
                    regrow = slot.dev is None or slot.dev.numel() < n or slot.dev.dtype != x.dtype
                    if (regrow and self.max_bytes is not None and self.slots_made > 2
                            and self.alloc_bytes - old_b + grown_b > self.max_bytes):
                        with torch.cuda.stream(self.copy_stream):
                            if slot.consumed is not None:
                                self.copy_stream.wait_event(slot.consumed)
                            slot.dev = None
                        self.alloc_bytes -= old_b
                        self.slots_made -= 1
                        self.peak_bytes = max(self.peak_bytes, self.alloc_bytes)
                        continue
                    break
                if self._stop or slot is None:
                    break
                with torch.cuda.stream(self.copy_stream):
                    if slot.consumed is not None:
                        self.copy_stream.wait_event(slot.consumed)   # kernels that read this slot are done
                    if slot.dev is None or slot.dev.numel() < n or slot.dev.dtype != x.dtype:
                        old = 0 if slot.dev is None else slot.dev.numel() * slot.dev.element_size()
                        slot.dev = None                      # (the old buffer goes back to the allocator before the new one is taken)
                        slot.dev = torch.empty(int(n * 1.25), dtype=x.dtype, device=self.device)
                        self.alloc_bytes += slot.dev.numel() * slot.dev.element_size() - old
                        self.peak_bytes = max(self.peak_bytes, self.alloc_bytes)
                        slot.copied = torch.cuda.Event()
                    slot.dev[:n].copy_(x.reshape(-1), non_blocking=True)
                    slot.copied.record(self.copy_stream)
                self._ready.put((slot, slot.dev[:n].view(x.shape), int(item["label"]), i))
        except Exception as e:                               # surfaced in the consumer thread
            self._ready.put(("error", e))
        self._ready.put(None)

    def __iter__(self) -> Iterator[Dict]:
        compute = torch.cuda.current_stream(self.device) if self.cuda else None
        prev = None
        try:
            while True:
                got = self._ready.get()
                if prev is not None or not self.cuda:        # hand the previous slot back to the copy thread
                    if self.cuda:
                        ev = torch.cuda.Event()
                        ev.record(compute)
                        prev.consumed = ev
                    if got is not None or self.cuda:
                        self._free.put(prev)
                    prev = None
                if got is None:
                    return
                if got[0] == "error":
                    raise got[1]
                if got[0] == "flush":                        # (only groups care; the previous slot has just been handed back)
                    continue
                slot, view, label, i = got
                if self.cuda:
                    compute.wait_event(slot.copied)
                    prev = slot
                yield {"input": view, "label": label, "index": i}
        finally:
            self.close()


    def iter_groups(self, group: int) -> Iterator[list]:
        """Yield LISTS of up to `group` items (same dicts as __iter__) whose ring slots all stay valid until the next list is
        requested: what a batched launch over several bags needs (acmil_ga_forward_batch).  Needs depth >= 2 * group for the
        copy of the next group to overlap the compute of this one (staged_groups sizes it).  A group is SHORTER than `group` when the
        ring's byte budget is reached first (large slides): the reader's "flush" marker closes it."""
        compute = torch.cuda.current_stream(self.device) if self.cuda else None
        held: list = []
        done = False
        try:
            while not done:
                if held:                                     # the previous group's kernels are enqueued: release its slots
                    if self.cuda:
                        ev = torch.cuda.Event()
                        ev.record(compute)
                        for sl in held:
                            sl.consumed = ev
                    for sl in held:
                        self._free.put(sl)
                    held = []
                items = []
                while len(items) < group:
                    got = self._ready.get()
                    if got is None:
                        done = True
                        break
                    if got[0] == "error":
                        raise got[1]
                    if got[0] == "flush":                    # the ring is full: launch what is staged (its slots come back after that)
                        if items:
                            break
                        continue
                    slot, view, label, i = got
                    if self.cuda:
                        compute.wait_event(slot.copied)
                    held.append(slot)
                    items.append({"input": view, "label": label, "index": i})
                if items:
                    yield items
        finally:
            self.close()


def staged(dataset, order: Iterable[int], device, depth: int = 3) -> BagPrefetcher:
    return BagPrefetcher(dataset, list(order), torch.device(device), depth=depth)


STAGE_RING_BYTES = 16 << 30      # ceiling of the staging ring of a grouped loop (also capped at 1/8 of the device's memory)


def staged_groups(dataset, order: Iterable[int], device, group: int = 16, max_bytes: Optional[int] = None) -> Iterator[list]:
    """Groups of up to `group` staged bags (for one batched launch each).  The ring holds up to two groups, and never more than
    `max_bytes` of device memory (default: min(16 GiB, 1/8 of the device)): with 2 x 64 slots each grown to the largest bag seen, a
    set of 100 000 x 512 fp32 slides would otherwise pin 33 GB for staging alone; groups get shorter instead."""
    device = torch.device(device)
    if max_bytes is None and device.type == "cuda":
        max_bytes = min(STAGE_RING_BYTES, torch.cuda.get_device_properties(device).total_memory // 8)
    return BagPrefetcher(dataset, list(order), device, depth=2 * group, max_bytes=max_bytes).iter_groups(group)


class TrainGroupPrefetcher:
    """Groups of `group` consecutive bags of `order` for the GROUP training step (ACMIL_GA.train_step_batch): the bags of a group are
    copied H2D straight into the row ranges of ONE ring slot, so the step gets its rows back to back without a device-side concatenation.
    Yields {'input': [sum N_b, D] device tensor (stored dtype), 'rows': [N_b], 'labels': [int], 'indices': [i]}; the view aliases a ring
    slot and is valid until the next group is requested.  Same event discipline as BagPrefetcher (copy stream, copied / consumed events);
    the last group of an epoch may be shorter.  Bags of one group that differ in dtype are widened to fp32 on the host (exact)."""

    def __init__(self, dataset, order: Sequence[int], device: torch.device, group: int, depth: int = 3):
        self.dataset, self.order, self.device = dataset, list(order), torch.device(device)
        self.group = max(1, int(group))
        self.depth = max(2, depth)
        self.cuda = self.device.type == "cuda"
        self._ready: "queue.Queue" = queue.Queue()
        self._free: "queue.Queue" = queue.Queue()
        self._stop = False
        for _ in range(self.depth):
            self._free.put(_Slot() if self.cuda else None)
        if self.cuda:
            self.copy_stream = torch.cuda.Stream(device=self.device)
        self._thread = threading.Thread(target=self._reader, daemon=True)
        self._thread.start()

    def __len__(self):
        return (len(self.order) + self.group - 1) // self.group

    def close(self):
        self._stop = True
        self._free.put(None)

    def _reader(self):
        try:
            if self.cuda:
                torch.cuda.set_device(self.device)
            for g0 in range(0, len(self.order), self.group):
                idx = self.order[g0:g0 + self.group]
                items = [self.dataset[i] for i in idx]
                xs = [torch.as_tensor(it["input"]) for it in items]
                if any(x.dtype != xs[0].dtype for x in xs):
                    xs = [x.float() for x in xs]
                rows = [int(x.shape[0]) for x in xs]
                labels = [int(it["label"]) for it in items]
                d = int(xs[0].shape[1])
                slot = self._free.get()
                if self._stop:
                    break
                if not self.cuda:
                    self._ready.put((None, torch.cat(xs, 0) if len(xs) > 1 else xs[0], rows, labels, idx))
                    continue
                n = sum(rows) * d
                with torch.cuda.stream(self.copy_stream):
                    if slot.consumed is not None:
                        self.copy_stream.wait_event(slot.consumed)
                    if slot.dev is None or slot.dev.numel() < n or slot.dev.dtype != xs[0].dtype:
                        slot.dev = None
                        slot.dev = torch.empty(int(n * 1.25), dtype=xs[0].dtype, device=self.device)
                        slot.copied = torch.cuda.Event()
                    off = 0
                    for x, r in zip(xs, rows):
                        slot.dev[off:off + r * d].copy_(x.reshape(-1), non_blocking=True)
                        off += r * d
                    slot.copied.record(self.copy_stream)
                self._ready.put((slot, slot.dev[:n].view(sum(rows), d), rows, labels, idx))
        except Exception as e:
            self._ready.put(("error", e))
        self._ready.put(None)

    def __iter__(self) -> Iterator[Dict]:
        compute = torch.cuda.current_stream(self.device) if self.cuda else None
        prev = None
        first = True
        try:
            while True:
                got = self._ready.get()
                if not first:
                    if self.cuda and prev is not None:
                        ev = torch.cuda.Event()
                        ev.record(compute)
                        prev.consumed = ev
                    self._free.put(prev)
                    prev = None
                if got is None:
                    return
                if got[0] == "error":
                    raise got[1]
                slot, view, rows, labels, idx = got
                if self.cuda:
                    compute.wait_event(slot.copied)
                prev = slot
                first = False
                yield {"input": view, "rows": rows, "labels": labels, "indices": list(idx)}
        finally:
            self.close()


def staged_train_groups(dataset, order: Iterable[int], device, group: int, depth: int = 3) -> TrainGroupPrefetcher:
    return TrainGroupPrefetcher(dataset, list(order), torch.device(device), group, depth=depth)
