"""Practical HBM write ceiling on this box: hipMemset of a large buffer, HIP-event timed (calibrates the
write-bound kernels of DESIGN.md section 3)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from revrand_amd import _hip
dev = _hip.get_device()
nbytes = 16 << 30
buf = dev.malloc(nbytes)
for rep in range(4):
    dev.timer_start()
    dev.memset(buf)
    ms = dev.timer_stop()
print("hipMemset of %d GiB: %.2f ms = %.2f TB/s" % (nbytes >> 30, ms, nbytes / ms / 1e9))
buf.free()
