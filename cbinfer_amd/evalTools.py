"""Evaluation harness with the reference's protocol (poseDetection/evalTools.py, symlinked into
sceneLabeling/): run a model over a frame list, time the LAST frame after priming on the others
(min of 3 repeats), list the change-based layers, write result tables.  Power logging, Tegra-specific in
the reference (tx2power.py), reads the amdgpu hwmon sensor here."""
import csv
import glob
import math
import os
import threading
import time

import torch

from . import CBConv2d, CBPoolMax2d, clearMemory


def _staged(frameset, cuda, preprocessor):
    """The frame list after the optional per-frame preprocessor, resident on the device if asked."""
    frames = [preprocessor(f) for f in frameset] if preprocessor is not None else list(frameset)
    return [f.cuda() for f in frames] if cuda else frames


def inferFramesetBenchmark(m, frameset, cuda=True, numIter=3, preprocessor=None):
    """Seconds for the last frame of `frameset` (protocol of the reference's evalTools.py:7-35): state
    cleared, frames [:-1] run untimed, device synchronised, then the last frame + synchronise is timed;
    repeated numIter times, minimum returned.  Frames are moved to the device before timing."""
    frames = _staged(frameset, cuda, preprocessor)
    wait = torch.cuda.synchronize if cuda else (lambda: None)
    best = float('inf')
    with torch.no_grad():
        for _ in range(numIter):
            clearMemory(m)
            for frame in frames[:-1]:
                m(frame)
            wait()
            t0 = time.perf_counter()
            m(frames[-1])
            wait()
            best = min(best, time.perf_counter() - t0)
    return best


def inferFrameset(m, frameset, cuda=True, preprocessor=None, postproc=None):
    """Output for the last frame after feeding the whole list (reference: evalTools.py:37-49)."""
    frameset = _staged(frameset, cuda, preprocessor)
    clearMemory(m)
    y = None
    with torch.no_grad():
        for frame in frameset:
            y = m(frame)
    if postproc is not None:
        y = postproc(y)
    return y


def _power_sensor(device=0):
    """sysfs file with the board power of HIP device `device` in microwatts: the amdgpu hwmon node
    (`power1_average` on older parts, `power1_input` on MI300/MI355X) of the card whose PCI address is the
    device's; the first card with such a node if the address cannot be matched; None if there is none."""
    nodes = []
    for name in ('power1_average', 'power1_input'):
        nodes += glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*/' + name)
    nodes = sorted(nodes)
    if not nodes:
        return None
    try:
        prop = torch.cuda.get_device_properties(device)
        addr = '%04x:%02x:%02x.' % (prop.pci_domain_id, prop.pci_bus_id, prop.pci_device_id)
        for n in nodes:
            card = os.path.realpath(os.path.join(os.path.dirname(n), '..', '..'))
            if os.path.basename(card).startswith(addr):
                return n
    except Exception:
        pass
    return nodes[0]


class PowerLogger(object):
    """Samples the GPU board power in a thread (the reference's tx2power.PowerLogger reads the Tegra
    INA3221 rails, sceneLabeling/tx2power.py; here the amdgpu hwmon sensor, see _power_sensor)."""

    def __init__(self, interval=0.05, device=0):
        self.interval = interval
        self.path = _power_sensor(device)
        self.samples = []     # (time, watts)
        self.events = []
        self._stop = threading.Event()
        self._thread = None

    def _read(self):
        try:
            with open(self.path) as f:
                return int(f.read().strip()) * 1e-6
        except Exception:
            return float('nan')

    def _run(self):
        while not self._stop.is_set():
            self.samples.append((time.time(), self._read()))
            time.sleep(self.interval)

    def start(self):
        self._stop.clear()
        self._thread = threading.Thread(target=self._run, daemon=True)
        self._thread.start()

    def recordEvent(self, name):
        self.events.append((time.time(), name))

    def stop(self):
        self._stop.set()
        if self._thread is not None:
            self._thread.join()
        self.samples.append((time.time(), self._read()))      # closing sample: the run ends here

    def getTotalEnergy(self):
        """Joules, trapezoidal over the samples."""
        e = 0.0
        for (t0, p0), (t1, p1) in zip(self.samples[:-1], self.samples[1:]):
            if not (math.isnan(p0) or math.isnan(p1)):
                e += 0.5 * (p0 + p1) * (t1 - t0)
        return e

    def getAveragePower(self):
        vals = [p for _, p in self.samples if not math.isnan(p)]
        return sum(vals) / len(vals) if vals else float('nan')


def inferFramesetPowerMeasurement(m, frameset, cuda=True, numFrames=0, preprocessor=None, interval=0.05):
    """Run a (back-and-forth extended) sequence under a PowerLogger (reference: evalTools.py:54-83)."""
    frameset = _staged(frameset, cuda, preprocessor)
    if numFrames > 0:
        frameset = frameset + frameset[-2:0:-1]
        frameset = int(math.ceil(numFrames / float(len(frameset)))) * frameset
        frameset = frameset[:numFrames]
    clearMemory(m)
    with torch.no_grad():
        m(frameset[0])
        if cuda:
            torch.cuda.synchronize()
        pl = PowerLogger(interval=interval)
        pl.start()
        for frame in frameset[1:]:
            m(frame)
        if cuda:
            torch.cuda.synchronize()
        pl.stop()
    return pl


def _submodels(model):
    return [m for _, m in sorted(model.named_children(), key=lambda kv: kv[0])]


def getCBconvLayers(model):
    """CBConv2d modules of a two-level model (sub-models sorted by name), reference: evalTools.py:85-93."""
    out = []
    for sub in _submodels(model):
        mods = sub if isinstance(sub, torch.nn.Sequential) else [sub]
        for m in mods:
            if type(m) is CBConv2d:
                out.append(m)
    return out


def getCBpoolLayers(model):
    out = []
    for sub in _submodels(model):
        mods = sub if isinstance(sub, torch.nn.Sequential) else [sub]
        for m in mods:
            if type(m) is CBPoolMax2d:
                out.append(m)
    return out


def writeTable(table, appName='poseDet', seqName='sampleSequence', evalName='evalXX', resultsDir='./results'):
    """Write a list of rows (first row = header) as ./results/<app>-<seq>-<eval>.csv
    (reference: evalTools.py:105-125)."""
    os.makedirs(resultsDir, exist_ok=True)
    path = os.path.join(resultsDir, '%s-%s-%s.csv' % (appName, seqName, evalName))
    with open(path, 'w', newline='') as f:
        w = csv.writer(f)
        for row in table:
            w.writerow(row)
    return path
