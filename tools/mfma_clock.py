import sys; sys.path.insert(0, ".")
from humanoid_amp_amd import _native as nat
for it in (64, 256, 1024):
    print(it, "constant", nat.calibrate_mfma_f16(False, it, with_clock=True), "random", nat.calibrate_mfma_f16(True, it, with_clock=True))
