"""`box`: what the GPU ran at during the timed region (clocks, power) and the class of the box from four calibration loops."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from audiogpt_amd import config as C            # noqa: E402
from audiogpt_amd import weights as WT          # noqa: E402


class BoxSampler:
    """Best-effort record of what the GPU ran at during the timed region: a thread reads the amdgpu sysfs nodes of one card
    (current shader clock level of pp_dpm_sclk, socket power of its hwmon) twice a second.  Box-to-box spread of one binary is
    several percent and follows the clock a box sustains under this load; nothing here is required --
    every failure yields None."""

    def __init__(self, device=None, root="/sys/class/drm", period=0.5, pci_root="/sys/bus/pci/devices"):
        """device: the torch device the benchmark runs on.  Its PCI address (domain:bus:device.function of the HIP device, from
        torch.cuda.get_device_properties / hipDeviceGetPCIBusId) selects the sysfs node; only when that cannot be resolved does
        the sampler fall back to the first amdgpu card it finds, and says so in `matched_by`."""
        import glob
        import threading
        self.sclk, self.power, self.period = [], [], period
        self.other = {"mclk": [], "fclk": [], "socclk": []}      # memory / fabric / SoC clock levels, where the driver exposes them
        self._stop = threading.Event()
        self._thread = None
        self.card = None
        self.bdf = self.pci_bdf(device)
        self.matched_by = None
        if self.bdf:
            cand = os.path.join(pci_root, self.bdf)
            if os.path.exists(os.path.join(cand, "pp_dpm_sclk")):
                self.card, self.matched_by = cand, "pci_bus_id"
            else:       # the same device through its DRM node (containers that hide /sys/bus/pci)
                for c in sorted(glob.glob(os.path.join(root, "card[0-9]*"))):
                    try:
                        if os.path.basename(os.path.realpath(os.path.join(c, "device"))) == self.bdf and \
                                os.path.exists(os.path.join(c, "device", "pp_dpm_sclk")):
                            self.card, self.matched_by = os.path.join(c, "device"), "drm_node_of_pci_bus_id"
                            break
                    except OSError:
                        continue
        if self.card is None:
            for c in sorted(glob.glob(os.path.join(root, "card[0-9]*"))):
                if os.path.exists(os.path.join(c, "device", "pp_dpm_sclk")):
                    self.card, self.matched_by = os.path.join(c, "device"), "first_amdgpu_card (PCI address of the HIP device not resolved)"
                    break
        self._hw = sorted(glob.glob(os.path.join(self.card, "hwmon", "hwmon*"))) if self.card else []

    @staticmethod
    def pci_bdf(device):
        """'dddd:bb:dd.f' of a torch CUDA(HIP) device, or None."""
        if device is None:
            return None
        try:
            pr = torch.cuda.get_device_properties(device)
            dom, bus, dv = (getattr(pr, k, None) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id"))
            if bus is not None and dv is not None:
                return "%04x:%02x:%02x.0" % (int(dom or 0), int(bus), int(dv))
        except Exception:
            pass
        try:      # older torch: ask the HIP runtime
            import ctypes
            hip = ctypes.CDLL("libamdhip64.so")
            buf = ctypes.create_string_buffer(64)
            idx = device.index if getattr(device, "index", None) is not None else torch.cuda.current_device()
            if hip.hipDeviceGetPCIBusId(buf, 64, int(idx)) == 0:
                return buf.value.decode().lower()
        except Exception:
            pass
        return None

    @staticmethod
    def parse_sclk(text):
        """MHz of the level pp_dpm_sclk marks with '*' (None if there is none)."""
        import re
        for line in text.splitlines():
            if line.rstrip().endswith("*"):
                m = re.search(r"(\d+)\s*mhz", line.lower())
                if m:
                    return int(m.group(1))
        return None

    def sample(self):
        try:
            v = self.parse_sclk(open(os.path.join(self.card, "pp_dpm_sclk")).read())
            if v is not None:
                self.sclk.append(v)
        except Exception:
            pass
        for k, v in self.other.items():
            try:
                c = self.parse_sclk(open(os.path.join(self.card, "pp_dpm_" + k)).read())
                if c is not None:
                    v.append(c)
            except Exception:
                pass
        for h in self._hw:
            for f in ("power1_average", "power1_input"):
                try:
                    self.power.append(int(open(os.path.join(h, f)).read().strip()) / 1e6)     # microwatts
                    return
                except Exception:
                    continue

    def __enter__(self):
        import threading
        if self.card:
            def loop():
                while not self._stop.wait(self.period):
                    self.sample()
            self._thread = threading.Thread(target=loop, daemon=True)
            self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        if self._thread is not None:
            self._thread.join(timeout=2.0)
        return False

    def summary(self):
        def med(v):
            return sorted(v)[len(v) // 2] if v else None
        return {"sclk_mhz_median": med(self.sclk), "sclk_mhz_min": min(self.sclk) if self.sclk else None,
                "mclk_mhz_median": med(self.other["mclk"]), "fclk_mhz_median": med(self.other["fclk"]),
                "socclk_mhz_median": med(self.other["socclk"]),
                "socket_power_w_median": med(self.power), "socket_power_w_max": max(self.power) if self.power else None,
                "samples": len(self.sclk), "pci_bus_id": self.bdf, "matched_by": self.matched_by,
                "source": "amdgpu sysfs (pp_dpm_sclk / mclk / fclk / socclk, hwmon power) of %s, sampled during the timed region" % self.card}


# What the calibration reads reach on the boxes that gave the fast-class numbers (profiles/README.md: 23.4 TB/s out of L2,
# 6.5 - 6.6 TB/s out of the Infinity Cache, 5.2 - 5.4 TB/s copy).  A box is put in the slow class when a read falls below 85 % of
# that: the kernels that lose on such boxes are the L2 -> LDS-bound ones, which neither the MFMA loop nor the
# copy loop tells apart.
BOX_CLASS_REF = {"l2_read_gbs": 23400.0, "infinity_cache_read_gbs": 6500.0, "copy_gbs": 5200.0, "mfma_bf16_tflops": 2250.0}


def box_class(calib, frac=0.85):
    """'fast' or 'slow(<which reads are low>)' from box.calib; None if the calibration did not run."""
    if not isinstance(calib, dict) or "error" in calib:
        return None
    low = [k for k, ref in BOX_CLASS_REF.items() if isinstance(calib.get(k), (int, float)) and calib[k] < frac * ref]
    return "fast" if not low else "slow(%s)" % ",".join(low)
