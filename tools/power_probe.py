"""Board power (hwmon power1_input, uW: the socket's PPT) and shader clock (freq1_input, Hz) of ONE card under load,
sampled from sysfs by the host thread while the device works through queued launches.

Which card: the HIP device's PCI address (torch.cuda.get_device_properties(i).pci_domain_id / pci_bus_id / pci_device_id ->
/sys/bus/pci/devices/<dddd:bb:dd.f>/hwmon/hwmon*).  Only when that lookup is impossible does it fall back to "the card whose
power rises under the load", with the idle reading taken when the probe is CREATED (create it before any warm-up) and a
refusal when the rise is ambiguous (a second card rising by more than half as much: another tenant's load)."""
import glob
import os
import time


def _hwmon_of_device(device_index):
    try:
        import torch
        p = torch.cuda.get_device_properties(device_index)
        dom, bus, dev = getattr(p, "pci_domain_id", 0), p.pci_bus_id, p.pci_device_id
    except Exception:
        return None
    for fn in range(8):
        bdf = "%04x:%02x:%02x.%d" % (dom, bus, dev, fn)
        for d in sorted(glob.glob("/sys/bus/pci/devices/%s/hwmon/hwmon*" % bdf)):
            if os.path.exists(os.path.join(d, "power1_input")):
                return d
    return None


class PowerProbe:
    def __init__(self, device_index=0):
        self.cards, self.by = [], "pci"
        d = _hwmon_of_device(device_index)
        if d:
            self.cards = [d]
        else:
            self.by = "largest rise"
            for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
                if os.path.exists(os.path.join(d, "power1_input")):
                    self.cards.append(d)
        self.idle = self._read_all("power1_input")

    def _read_all(self, name):
        out = []
        for d in self.cards:
            try:
                out.append(float(open(os.path.join(d, name)).read().strip()))
            except (OSError, ValueError):
                out.append(float("nan"))
        return out

    def measure(self, busy, interval=0.02):
        """busy() -> bool: is the device still working?  Samples until it is not."""
        import numpy as np
        pw, fq = [], []
        while busy():
            pw.append(self._read_all("power1_input"))
            fq.append(self._read_all("freq1_input"))
            time.sleep(interval)
        if len(pw) < 8 or not self.cards:
            return None
        p, f = np.array(pw), np.array(fq)
        steady = p[(3 * len(p)) // 5:]                         # (the sensor is a running average about a second long: the
                                                               #  last two fifths of a three-second load are the settled part)
        c = 0
        if len(self.cards) > 1:
            rise = np.nanmean(steady, axis=0) - np.array(self.idle)
            order = np.argsort(-np.nan_to_num(rise, nan=-1e30))
            c = int(order[0])
            if rise[order[1]] > 0.5 * rise[c]:
                return {"error": "ambiguous: two cards rose under the load (%.0f W and %.0f W) and the device's PCI address "
                                 "could not be mapped to a hwmon directory" % (rise[c] / 1e6, rise[order[1]] / 1e6)}
        cap = None
        try:
            cap = float(open(os.path.join(self.cards[c], "power1_cap")).read()) / 1e6
        except (OSError, ValueError):
            pass
        return {"watts_avg": round(float(np.nanmean(steady[:, c])) / 1e6, 1), "watts_max": round(float(np.nanmax(p[:, c])) / 1e6, 1),
                "watts_before": round(self.idle[c] / 1e6, 1), "watts_cap": cap,
                "sclk_MHz_avg": round(float(np.nanmean(f[(3 * len(f)) // 5:, c])) / 1e6, 0), "samples": int(len(p)),
                "card": "%s (%s)" % (self.cards[c], self.by),
                "source": "hwmon power1_input (PPT) / freq1_input, %d ms apart" % int(interval * 1e3)}


def sample_load(step, seconds, ms_per_step, stream, device_index=0, probe=None):
    """About `seconds` of step() launches on `stream` (a torch stream) with the card sampled meanwhile.  The sampler runs on
    its own thread from before the first launch: queueing tens of thousands of small launches blocks the host once the
    runtime's queue is full, and a sampler that starts after the queueing would see only the tail of the load."""
    import threading
    probe = probe or PowerProbe(device_index)
    n = max(8, int(seconds / (ms_per_step * 1e-3)))
    running = [True]
    out = [None]

    def sampler():
        out[0] = probe.measure(lambda: running[0])
    t = threading.Thread(target=sampler)
    t.start()
    try:
        for _ in range(n):
            step()
        stream.synchronize()
    finally:
        running[0] = False
        t.join()
    pw = out[0]
    if pw and "error" not in pw:
        pw["launches_sampled"] = n
    return pw
