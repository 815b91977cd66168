import sys, torch
sys.path.insert(0, ".")
from humanoid_amp_amd import _native as nat
lib = nat.load()
N = 65536
g = torch.Generator(device="cuda").manual_seed(0)
mask = (torch.rand(N, generator=g, device="cuda") < 0.01)
ids = torch.empty(N, dtype=torch.int64, device="cuda"); count = torch.zeros(1, dtype=torch.int64, device="cuda")
for te in (64, 32, 16, 8):
    counts = mask.view(-1, te).sum(1).to(torch.int32).contiguous()
    for _ in range(5):
        nat.check(lib.amp_reset_compact_tiles(nat.dptr(mask), nat.dptr(counts), te, N, nat.dptr(ids), nat.dptr(count), nat.stream_ptr()), "x")
    torch.cuda.synchronize()
    with nat.KernelTrace(64) as t:
        for _ in range(20):
            nat.check(lib.amp_reset_compact_tiles(nat.dptr(mask), nat.dptr(counts), te, N, nat.dptr(ids), nat.dptr(count), nat.stream_ptr()), "x")
    r = [ms for _, ms in t.records()]
    ok = torch.equal(ids[: int(count)], mask.nonzero().squeeze(-1))
    print("tile_envs", te, "us", round(sum(r) / len(r) * 1e3, 2), "ok", ok)
