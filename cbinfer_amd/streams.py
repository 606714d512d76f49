"""Side streams that really run beside the caller's stream.

HIP maps streams onto a few hardware queues in creation order; two streams that land on the same queue take turns, and
what looks like a fork in the code runs serially (measured in one bench process: the two branches of OpenPose's stages on
"two streams" 998 frames/s against 1028 serial, in a fresh process 1260; frame pipelining 10.4k against 11.6k).  Which
queue a new stream gets depends on how many streams the process created before -- nothing a library can know.  So the
side stream is PROBED: two spin kernels, one on the caller's stream and one on the candidate, must take the time of
one; candidates that share the caller's queue are kept alive (destroying one would hand its slot to the next) and the
next one is tried.  A probe is six device-wide synchronisations around 4 x 10^5-cycle spin kernels (~0.2 ms each): a few
milliseconds per candidate, once per device and caller stream -- inside the first forward that asks, never in a capture.
The reference runs on one stream."""
import time
import warnings

import torch

_KEEP_MAX = 32   # candidates kept alive at most (beyond that the oldest go: their queue slots are long since taken)
_kept = []       # candidates created while probing (see above)
_found = {}      # (device index, raw handle of the caller's stream) -> (caller's stream object, side stream)


def _keep(stream):
    _kept.append(stream)
    if len(_kept) > _KEEP_MAX:
        del _kept[:len(_kept) - _KEEP_MAX]


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


def overlaps(a, b, cycles=400000):
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
        hit = _found.get(key)
        # (raw handles are recycled: the entry counts only while the stream object it was probed against is the caller's)
        if hit is not None and hit[0] == cur:
            return hit[1]
        if torch.cuda.is_current_stream_capturing() or not hasattr(torch.cuda, '_sleep'):
            return torch.cuda.Stream()             # (no probing inside a capture, and nothing cached)
        for _ in range(tries):
            cand = torch.cuda.Stream()
            _keep(cand)
            if overlaps(cur, cand):
                _found[key] = (cur, cand)
                return cand
        # no candidate overlapped: a fresh stream, NOT cached (the next call probes again), and said aloud
        warnings.warn("cbinfer_amd.streams.side_stream: none of %d candidate streams overlaps with the current stream; "
                      "forked work will run serially" % tries, RuntimeWarning, stacklevel=2)
        return torch.cuda.Stream()


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
        cur = torch.cuda.current_stream()      # (the chosen streams must overlap with the caller's stream as well)
        for _ in range(tries):
            if len(chosen) == n:
                break
            cand = torch.cuda.Stream()
            _keep(cand)
            if all(overlaps(c, cand) for c in [cur] + chosen):
                chosen.append(cand)
            else:
                spare.append(cand)
        return (chosen + spare + [torch.cuda.Stream() for _ in range(n)])[:n]
