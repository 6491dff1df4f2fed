"""Prints the device's peak shader clock and the frequency of wall_clock64() (s_memrealtime), which the phase-timing
diagnostics use as their time base."""
import ctypes

import torch  # noqa: F401  (loads libamdhip64)

hip = ctypes.CDLL("libamdhip64.so")
v = ctypes.c_int()
for name, attr in (("hipDeviceAttributeClockRate (kHz)", 5), ("hipDeviceAttributeWallClockRate (kHz)", 10017),
                   ("hipDeviceAttributeMemoryClockRate (kHz)", 60)):
    rc = hip.hipDeviceGetAttribute(ctypes.byref(v), attr, 0)
    print(name, v.value if rc == 0 else f"error {rc}")
