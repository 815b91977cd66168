"""What clock / rate do the matrix pipes get in SHORT bursts between memory-bound phases?  (Layer 1's k-loop is a ~13-us MFMA
burst at ~50 % pipe occupancy between ~10-us store phases; would a denser k-loop keep its clock?)  Enqueues, without host
syncs, [a 134-MB fill (~25 us, memory-bound) | a bare-MFMA burst of iters x 48 MFMAs per wave on random operands] x 24 and
reports the bursts' TFLOP/s (engine tracer) and the core clock inside the last one."""
import ctypes as C
import sys

import torch

sys.path.insert(0, ".")
from humanoid_amp_amd import _native as nat  # noqa: E402

lib = nat.load()
dev = torch.device("cuda:0")
cus = torch.cuda.get_device_properties(dev).multi_processor_count
scratch = torch.zeros(cus * 256 + cus * 4, dtype=torch.float32, device=dev)
filler = torch.empty(134 * 1024 * 1024 // 4, dtype=torch.float32, device=dev)
flops = C.c_double()
for random_ops in (True, False):
    for iters in (4, 8, 16, 32, 64, 256):
        for gap in (True, False):
            with nat.KernelTrace(capacity=64, kernel_filter="mfma_f16_calibration_kernel") as tr:
                for _ in range(24):
                    if gap:
                        filler.fill_(1.0)
                    nat.check(lib.amp_calibrate_mfma_f16(int(random_ops), iters, nat.dptr(scratch), scratch.numel(), C.byref(flops),
                                                         nat.stream_ptr()), "amp_calibrate_mfma_f16")
            torch.cuda.synchronize()
            ms = sorted(r[1] for r in tr.records()[4:])
            med = ms[len(ms) // 2]
            ticks = scratch[cus * 256:].view(torch.int64).view(cus, 2).double().cpu()
            mhz = float((ticks[:, 0] / ticks[:, 1].clamp(min=1)).median()) * 100.0
            print(f"{'random  ' if random_ops else 'constant'} operands, {iters:4d} x 48 MFMAs per wave, {'fill between' if gap else 'back to back'}: "
                  f"burst {med * 1e3:7.1f} us  {flops.value / (med * 1e-3) / 1e12:7.1f} TFLOP/s  core clock {mhz:6.0f} MHz")
