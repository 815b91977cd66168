"""Oracle: Philox4x32-10 + the device-side sample_times draw (TEST INFRASTRUCTURE ONLY).

The reference samples reset times with the host's global numpy RNG (motions/motion_loader.py:321-327), which cannot be
reproduced on a device; the engine's device reset path uses a counter-based generator instead (parity with the
reference: distributional only).  This numpy restatement of the published Philox4x32-10 algorithm (Salmon et al.,
"Parallel Random Numbers: As Easy as 1, 2, 3", SC'11; constants as in Random123) pins the HIP kernel bit for bit, and
is itself pinned by the Random123 known-answer vectors in tests/test_oracle_rng.py.
"""

from __future__ import annotations

import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised over numpy uint64 arrays holding 32-bit words; returns four uint64 arrays of 32-bit words."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) & MASK for c in (c0, c1, c2, c3))
    k0, k1 = int(k0) & 0xFFFFFFFF, int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2
        n0 = (p1 >> np.uint64(32)) ^ c1 ^ np.uint64(k0)
        n1 = p1 & MASK
        n2 = (p0 >> np.uint64(32)) ^ c3 ^ np.uint64(k1)
        n3 = p0 & MASK
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return c0, c1, c2, c3


def sample_times(durations: np.ndarray, seed: int, step: int, index: np.ndarray, start: bool = False):
    """(motion_ids int64, times float64) exactly as csrc/motion.hip::sample_times_kernel draws them."""
    index = np.asarray(index, dtype=np.uint64)
    n_clips = len(durations)
    r0, r1, r2, _ = philox4x32_10(index & MASK, index >> np.uint64(32), step & 0xFFFFFFFF, (step >> 32) & 0xFFFFFFFF,
                                  seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    ids = ((r0 * np.uint64(n_clips)) >> np.uint64(32)).astype(np.int64)
    u = ((r1 >> np.uint64(5)).astype(np.float64) * 67108864.0 + (r2 >> np.uint64(6)).astype(np.float64)) / 9007199254740992.0
    times = np.zeros(len(index)) if start else u * np.asarray(durations, dtype=np.float64)[ids]
    return ids, times


def ring_sample_indices(size: int, seed: int, draw: int, n: int, first_row: int = 0) -> np.ndarray:
    """Storage rows csrc/ring.hip::ring_sample_kernel draws: floor(word0(Philox(counter=(first_row + i, draw), key=seed)) * size / 2^32)."""
    i = np.arange(n, dtype=np.uint64) + np.uint64(first_row)
    r0, _, _, _ = philox4x32_10(i & MASK, i >> np.uint64(32), draw & 0xFFFFFFFF, (draw >> 32) & 0xFFFFFFFF,
                                seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    return ((r0 * np.uint64(size)) >> np.uint64(32)).astype(np.int64)


def feistel_permutation(n: int, seed: int, epoch: int, positions: np.ndarray) -> np.ndarray:
    """pi(positions) of csrc/ring.hip::feistel_permute: the point-wise pseudo-random permutation of [0, n) keyed by (seed, epoch)."""
    bits = 1
    while (1 << bits) < n:
        bits += 1
    hb = (bits + 1) // 2
    mask = np.uint64((1 << hb) - 1)
    M32 = np.uint64(0xFFFFFFFF)
    v = np.asarray(positions, dtype=np.uint64).copy()
    todo = np.ones(v.shape, dtype=bool)
    first = True
    while todo.any():
        x = v[todo]
        L, R = (x >> np.uint64(hb)) & mask, x & mask
        for rd in range(6):
            k0 = np.uint64((seed + 0x9E3779B9 * (rd + 1)) & 0xFFFFFFFF)
            k1 = np.uint64(((seed >> 32) ^ epoch ^ ((0xBB67AE85 * (rd + 1)) & 0xFFFFFFFF) ^ (epoch >> 32)) & 0xFFFFFFFF)
            p = ((R ^ k0) & M32) * np.uint64(0xD2511F53)
            f = ((p >> np.uint64(32)) ^ (p & M32) ^ k1) & M32
            L, R = R, (L ^ (f & mask)) & mask
        x = (L << np.uint64(hb)) | R
        v[todo] = x
        todo = v >= np.uint64(n)
        first = False
    return v.astype(np.int64)


class RingOracle:
    """numpy restatement of skrl RandomMemory.add_samples on one tensor (write head wraps; oldest rows overwritten)."""

    def __init__(self, capacity: int, dim: int):
        self.rows = np.zeros((capacity, dim), dtype=np.float32)
        self.capacity, self.head, self.size = capacity, 0, 0

    def add(self, batch: np.ndarray) -> None:
        for row in batch:
            self.rows[self.head] = row
            self.head = (self.head + 1) % self.capacity
            self.size = min(self.size + 1, self.capacity)


COMMAND_DOMAIN = 0xA14C0000  # csrc/command.hip::kCommandDomain


def command_draw(seed: int, step: int, env: np.ndarray, mode: int, vel_lo: float, vel_span: float, t_lo: float, t_span: float):
    """(command [n, 2] float32, time_left [n] float32) exactly as csrc/command.hip::draw_command draws them for the
    GLOBAL env ids ``env``: Philox4x32-10(counter = (env, step), key = (seed_lo, seed_hi ^ (COMMAND_DOMAIN + mode))),
    u = (word >> 8) * 2^-24 in fp32, then the reference's ``torch.rand(...) * (hi - lo) + lo`` as one fp32 multiply and
    one fp32 add (g1_amp_env.py:152-166,421-434).  mode 0 = timer expiry, 1 = reset."""
    env = np.asarray(env, dtype=np.uint64)
    k1 = ((seed >> 32) & 0xFFFFFFFF) ^ ((COMMAND_DOMAIN + mode) & 0xFFFFFFFF)
    r0, r1, r2, _ = philox4x32_10(env & MASK, env >> np.uint64(32), step & 0xFFFFFFFF, (step >> 32) & 0xFFFFFFFF,
                                  seed & 0xFFFFFFFF, k1)
    f = np.float32
    u = [(r >> np.uint64(8)).astype(np.float32) * f(2.0 ** -24) for r in (r0, r1, r2)]
    cx = u[0] * f(vel_span) + f(vel_lo)
    cy = u[1] * f(vel_span) + f(vel_lo)
    tl = u[2] * f(t_span) + f(t_lo)
    return np.stack([cx, cy], axis=1).astype(np.float32), tl.astype(np.float32)


def command_tick(command: np.ndarray, time_left: np.ndarray, step_dt: float, vel_range, time_range, seed: int, step: int,
                 env_offset: int = 0):
    """G1AmpEnv._pre_physics_step's timer update (g1_amp_env.py:146-167) with the engine's counter-based draw in place
    of torch.rand: returns new (command, time_left)."""
    command, time_left = command.astype(np.float32).copy(), time_left.astype(np.float32).copy()
    time_left = (time_left - np.float32(step_dt)).astype(np.float32)
    lo, hi = float(vel_range[0]), float(vel_range[1])
    expired = np.nonzero(time_left <= 0.0)[0]
    if len(expired) > 0 and hi > lo:
        c, t = command_draw(seed, step, expired + env_offset, 0, lo, hi - lo, float(time_range[0]),
                            float(time_range[1]) - float(time_range[0]))
        command[expired], time_left[expired] = c, t
    return command, time_left


def command_reset(command: np.ndarray, time_left: np.ndarray, env_ids: np.ndarray, vel_range, time_range, seed: int, step: int,
                  env_offset: int = 0):
    """The command resample of G1AmpEnv._reset_strategy_random (g1_amp_env.py:421-439) for ``env_ids``."""
    command, time_left = command.astype(np.float32).copy(), time_left.astype(np.float32).copy()
    lo, hi = float(vel_range[0]), float(vel_range[1])
    env_ids = np.asarray(env_ids, dtype=np.int64)
    if hi > lo:
        c, t = command_draw(seed, step, env_ids + env_offset, 1, lo, hi - lo, float(time_range[0]),
                            float(time_range[1]) - float(time_range[0]))
        command[env_ids], time_left[env_ids] = c, t
    else:
        command[env_ids, 0], command[env_ids, 1], time_left[env_ids] = lo, 0.0, np.inf
    return command, time_left
