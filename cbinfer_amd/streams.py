"""Side streams that really run beside the caller's stream.

HIP maps streams onto a few hardware queues in creation order; two streams that land on the same queue take turns, and
what looks like a fork in the code runs serially (measured in one bench process: the two branches of OpenPose's stages on
"two streams" 998 frames/s against 1028 serial, in a fresh process 1260; frame pipelining 10.4k against 11.6k).  Which
queue a new stream gets depends on how many streams the process created before -- nothing a library can know.  So the
side stream is PROBED: two spin kernels, one on the caller's stream and one on the candidate, must take the time of
one; candidates that share the caller's queue are kept alive (destroying one would hand its slot to the next) and the
next one is tried.  A few milliseconds, once per device and caller stream.  The reference runs on one stream."""
import time

import torch

_kept = []       # every candidate ever created (see above)
_found = {}      # (device index, raw handle of the caller's stream) -> side stream


def _spin_pair_seconds(a, b, cycles):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(a):
        torch.cuda._sleep(cycles)
    if b is not None:
        with torch.cuda.stream(b):
            torch.cuda._sleep(cycles)
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def overlaps(a, b, cycles=1000000):
    """True if a spin kernel on stream a and one on stream b run concurrently (best of three)."""
    one = min(_spin_pair_seconds(a, None, cycles) for _ in range(3))
    two = min(_spin_pair_seconds(a, b, cycles) for _ in range(3))
    return two < 1.5 * one


def side_stream(device=None, tries=8):
    """A stream on `device` whose kernels overlap with those of the device's CURRENT stream."""
    dev = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
    index = dev.index if dev.index is not None else torch.cuda.current_device()
    with torch.cuda.device(index):
        cur = torch.cuda.current_stream()
        key = (index, cur.cuda_stream)
        if key in _found:
            return _found[key]
        if torch.cuda.is_current_stream_capturing() or not hasattr(torch.cuda, '_sleep'):
            _found[key] = torch.cuda.Stream()      # (no probing inside a capture)
            return _found[key]
        cand = None
        for _ in range(tries):
            cand = torch.cuda.Stream()
            _kept.append(cand)
            if overlaps(cur, cand):
                break
        _found[key] = cand
        return cand


def overlapping_streams(n, device=None, tries=12):
    """n streams whose kernels overlap pairwise (as far as the device's hardware queues allow: a candidate that cannot be
    told apart from an already chosen stream is skipped, and after `tries` candidates the list is filled up with what
    there is)."""
    dev = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
    index = dev.index if dev.index is not None else torch.cuda.current_device()
    with torch.cuda.device(index):
        if torch.cuda.is_current_stream_capturing() or not hasattr(torch.cuda, '_sleep'):
            return [torch.cuda.Stream() for _ in range(n)]
        chosen, spare = [], []
        for _ in range(tries):
            if len(chosen) == n:
                break
            cand = torch.cuda.Stream()
            _kept.append(cand)
            if all(overlaps(c, cand) for c in chosen):
                chosen.append(cand)
            else:
                spare.append(cand)
        return (chosen + spare + [torch.cuda.Stream() for _ in range(n)])[:n]
