// EXPERIMENT (not product, not built): the env-step DMA tile body as PERSISTENT workgroups with two LDS tile buffers --
// the next tile's LDS-DMA loads fly while the current tile is reduced and stored, per-env values staged component-major
// through LDS as well.  Bit-identical to the product body (89 GPU tests), but SLOWER: 65 536 G1 envs 46.2 us (launch
// alone) / 65.6 us (behind a cache flush) against 36.5 / 53-57 us for one tile per workgroup; 16-env tiles 61.6 / 75.1 us.
// Two 37-KB buffers leave two workgroups per CU (8 waves), and an occupancy sweep of the product body (LDS padding:
// 4 / 3 / 2 / 1 workgroups per CU = 36.5 / 36.7 / 40.4 / 44.2 us) shows the launch is bound by instruction issue per CU,
// not by memory latency: there is nothing for the prefetch to hide, and the extra barrier phase costs.
// Kept for the record; see DESIGN.md "tried and rejected".
// LDS of the DMA body, in floats: two tile buffers (the next tile's loads fly while this one is computed and stored)
// and the per-workgroup constants.  Host and device agree through this one struct.
struct EnvDmaLds {
  int img, act, acc, la, cmd, pe, buf;  // offsets inside a tile buffer / its size
  int red, mu, dn, total;               // offsets of the shared part (after the two buffers) / everything
  int lim;                              // soft limits: inside the tile buffer when they are per env, else shared
  int n_pe;                             // per-env components staged component-major: [n_pe][T]
};
__host__ __device__ inline EnvDmaLds env_dma_lds(int T, int KD, int nd, int n_key, bool per_env_limits) {
  EnvDmaLds l;
  const int ndT = (T * nd + 3) & ~3;  // flat [T, nd] blocks keep 16-B aligned bases (T >= 8: every size is a multiple of 4)
  l.n_pe = 13 + 3 * n_key + 2;        // root_pos 3 | root_quat 4 | lin 3 | ang 3 | key bodies 3 each | episode_length lo, hi
  l.img = 0;
  l.act = T * KD;
  l.acc = l.act + ndT;
  l.la = l.acc + ndT;
  l.cmd = l.la + ndT;
  l.pe = l.cmd + 2 * T;
  l.buf = l.pe + l.n_pe * T;
  if (per_env_limits) {  // [T, 2*nd] rows at an odd pitch, part of the tile
    l.lim = l.buf;
    l.buf = (l.lim + T * (2 * nd + 1) + 3) & ~3;
  }
  l.red = 2 * l.buf;
  l.mu = l.red + 4 * T;
  l.dn = l.mu + ((KD + 3) & ~3);
  l.total = l.dn + ((KD + 3) & ~3);
  if (!per_env_limits) {  // one shared [2*nd] row, staged once
    l.lim = l.total;
    l.total += 2 * nd + 1;
  }
  return l;
}

// Persistent workgroup `wg` of `n_wgs` walks the whole tiles wg, wg + n_wgs, ... < n_tiles.
template <int T>
__device__ __forceinline__ void env_step_dma_tiles(const EnvPlan& p, const AmpSimState& st, const AmpEnvBuffers& bf,
                                                   int64_t N, int wg, int n_wgs, int n_tiles, float* smem) {
  const int D = p.D, nd = p.n_dof, KD = p.K * p.D, C = KD - D;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // the wave that does the per-env work rotates with the workgroup index: the workgroups of a CU put it on different SIMDs
  const int role = (wave + wg) & 3;
  const bool g1 = p.reward_mode == 1;
  const bool extra = p.use_last_actions;  // policy obs = [obs[:Db] | last_actions | command]
  const bool has_cmd = (extra || g1) && p.use_command;
  const bool per_env_limits = g1 && st.soft_limits_stride != 0;
  const bool fused = bf.disc_input != nullptr;
  const bool scaled = fused && bf.scaler_mean != nullptr;
  const int lim_row = 2 * nd + 1;
  const EnvDmaLds L = env_dma_lds(T, KD, nd, p.n_key, per_env_limits);
  float* s_red = smem + L.red;  // [4, T]    reward partial sums
  float* s_mu = smem + L.mu;    // [K*D]     scaler mean
  float* s_dn = smem + L.dn;    // [K*D]     scaler sqrt(var) + eps

  // ---- HBM -> LDS of one tile, no registers in between -------------------------------------------------------
  auto issue = [&](const int tile, float* b) {
    const int64_t tile_base = (int64_t)tile * T;
    float* s_img = b + L.img;
    const float* hist = bf.amp_obs_buffer + tile_base * KD;
#pragma unroll 1
    for (int r = wave; r < T; r += 4) {
      float* row = s_img + r * KD;
#pragma unroll 1
      for (int c0 = 0; c0 < C; c0 += 64)  // slot k + 1 <- old slot k (g1_amp_env.py:187-190)
        if (c0 + lane < C) dma4(hist + r * KD + c0 + lane, row + D + c0);
      const float* gp = st.joint_pos + (tile_base + r) * st.joint_pos_stride;
      const float* gv = st.joint_vel + (tile_base + r) * st.joint_vel_stride;
#pragma unroll 1
      for (int c0 = 0; c0 < nd; c0 += 64)
        if (c0 + lane < nd) {
          dma4(gp + c0 + lane, row + c0);
          dma4(gv + c0 + lane, row + nd + c0);
        }
    }
    const int n16 = T * nd / 4;  // T * nd is a multiple of 4 (T >= 8)
#pragma unroll 1
    for (int pc = wave * 64; pc < n16; pc += kBlock) {
      const int i = pc + lane;
      if (i < n16) {
        if (g1) {
          dma16(st.actions + tile_base * nd + 4 * i, b + L.act + 4 * pc);
          dma16(st.joint_acc + tile_base * nd + 4 * i, b + L.acc + 4 * pc);
        }
        if (extra) dma16(st.last_actions + tile_base * nd + 4 * i, b + L.la + 4 * pc);
      }
    }
    if (has_cmd && wave == 3 && lane < T / 2) dma16(st.command + tile_base * 2 + 4 * lane, b + L.cmd);
    if (per_env_limits) {
#pragma unroll 1
      for (int r = wave; r < T; r += 4) {
        const float* gl = st.soft_limits + (tile_base + r) * st.soft_limits_stride;
#pragma unroll 1
        for (int c0 = 0; c0 < 2 * nd; c0 += 64)
          if (c0 + lane < 2 * nd) dma4(gl + c0 + lane, b + L.lim + r * lim_row + c0);
      }
    }
    // per-env values, one env per lane, staged component-major [component][T]
    const int64_t env = tile_base + (lane < T ? lane : 0);
    const int nk3 = 3 * p.n_key;
#pragma unroll 1
    for (int ci = wave; ci < L.n_pe; ci += 4) {
      const float* g;
      if (ci < 3) g = st.root_pos + env * st.root_pos_stride + ci;
      else if (ci < 7) g = st.root_quat + env * st.root_quat_stride + (ci - 3);
      else if (ci < 10) g = st.root_lin_vel + env * st.root_lin_vel_stride + (ci - 7);
      else if (ci < 13) g = st.root_ang_vel + env * st.root_ang_vel_stride + (ci - 10);
      else if (ci < 13 + nk3) {
        const int k = (ci - 13) / 3, c = (ci - 13) - 3 * k;
        int body = st.key_body[0];  // select chain: a dynamic index into the by-value struct would go through scratch
#pragma unroll
        for (int q = 1; q < kMaxKey; ++q) body = k == q ? st.key_body[q] : body;
        g = st.body_pos + env * st.body_pos_stride + (int64_t)body * 3 + c;
      } else {
        g = reinterpret_cast<const float*>(st.episode_length + env) + (ci - 13 - nk3);  // int64: low, high dword
      }
      if (lane < T) dma4(g, b + L.pe + ci * T);
    }
  };

  // ---- once per workgroup: scaler statistics, soft limits, the first tile --------------------------------------
  if (wg < n_tiles) issue(wg, smem);
  if (scaled) {
#pragma unroll 1
    for (int pc = wave * 64; pc < KD; pc += kBlock)
      if (pc + lane < KD) {
        dma4(bf.scaler_mean + pc + lane, s_mu + pc);
        dma4(bf.scaler_den + pc + lane, s_dn + pc);
      }
  }
  if (g1 && !per_env_limits)
    for (int e = tid; e < 2 * nd; e += kBlock) smem[L.lim + e] = st.soft_limits[e];
  // ---- per-lane constants of the output walks ---------------------------------------------------------------
  const bool blocks = bf.disc_input_format == AMP_DISC_INPUT_F16_BLOCKS;
  const float s_x = bf.disc_plane_scale, clip = bf.scaler_clip;
  const int64_t pitch = bf.disc_input_stride;
  const int P = p.P, Db = p.Db;

  int it = 0;
#pragma unroll 1
  for (int tile = wg; tile < n_tiles; tile += n_wgs, ++it) {
    float* const b = smem + (it & 1) * L.buf;
    const int64_t tile_base = (int64_t)tile * T;
    float* const s_img = b + L.img;  // [T, K*D]  the tile's new AMP rows (== its span of the AMP buffer)
    const float* const s_act = b + L.act;
    const float* const s_acc = b + L.acc;
    const float* const pe = b + L.pe;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's LDS-DMA pieces of `tile` have landed (and its stores left)
    __syncthreads();                                   // everyone's have; the other buffer has been read out
    if (tile + n_wgs < n_tiles) issue(tile + n_wgs, smem + ((it & 1) ^ 1) * L.buf);

    // ---- per-env work (role 0, one env per lane) beside three of the four reward reductions (roles 1..3) -------
    const bool env_lane = role == 0 && lane < T;
    const int64_t env = tile_base + lane;
    int died = 0;
    float rq[4], rl[3];
    if (role == 0) {
      int reset_bit = 0;
      if (env_lane) {
        const float px = pe[0 * T + lane], py = pe[1 * T + lane], pz = pe[2 * T + lane];
        rq[0] = pe[3 * T + lane]; rq[1] = pe[4 * T + lane]; rq[2] = pe[5 * T + lane]; rq[3] = pe[6 * T + lane];
        rl[0] = pe[7 * T + lane]; rl[1] = pe[8 * T + lane]; rl[2] = pe[9 * T + lane];
        const float ax = pe[10 * T + lane], ay = pe[11 * T + lane], az = pe[12 * T + lane];
        const int nk3 = 3 * p.n_key;
        const uint32_t lo = __float_as_uint(pe[(13 + nk3) * T + lane]), hi = __float_as_uint(pe[(14 + nk3) * T + lane]);
        const int64_t ep_len = (int64_t)(((uint64_t)hi << 32) | lo);
        // g1_amp_env.py:321-330
        const int tout = ep_len >= p.max_episode_length - 1;
        died = p.early_termination ? (pz < p.termination_height) : 0;
        bf.died[env] = (uint8_t)died;
        bf.time_out[env] = (uint8_t)tout;
        reset_bit = died | tout;
        if (bf.reset_mask) bf.reset_mask[env] = (uint8_t)reset_bit;
        // compute_obs features that are not plain copies (g1_amp_env.py:545-555)
        const Quat q{rq[0], rq[1], rq[2], rq[3]};
        const Vec3 tg = quat_apply_ref(q, Vec3{1.0f, 0.0f, 0.0f});
        const Vec3 nm = quat_apply_ref(q, Vec3{0.0f, 0.0f, 1.0f});
        float* o = s_img + lane * KD + 2 * nd;
        o[0] = pz;
        o[1] = tg.x; o[2] = tg.y; o[3] = tg.z;
        o[4] = nm.x; o[5] = nm.y; o[6] = nm.z;
        o[7] = rl[0]; o[8] = rl[1]; o[9] = rl[2];
        o[10] = ax; o[11] = ay; o[12] = az;
#pragma unroll 1
        for (int k = 0; k < p.n_key; ++k) {
          o[13 + 3 * k + 0] = pe[(13 + 3 * k + 0) * T + lane] - px;
          o[13 + 3 * k + 1] = pe[(13 + 3 * k + 1) * T + lane] - py;
          o[13 + 3 * k + 2] = pe[(13 + 3 * k + 2) * T + lane] - pz;
        }
      }
      if (bf.reset_tile_counts) {
        const unsigned long long bits = __ballot(reset_bit);
        if (lane == 0) bf.reset_tile_counts[tile] = __popcll(bits);
      }
    } else if (g1 && lane < T) {
      // compute_rewards (g1_amp_env.py:564-606): sums over the DoFs of env `lane`; role 1 takes two of the four terms
      if (role == 1) {
        float acc = 0.0f;
        for (int j = 0; j < nd; ++j) { const float a = s_act[lane * nd + j]; acc += a * a; }
        s_red[0 * T + lane] = acc;
        acc = 0.0f;
        for (int j = 0; j < nd; ++j) { const float a = s_acc[lane * nd + j]; acc += a * a; }
        s_red[2 * T + lane] = acc;
      } else if (role == 2) {
        const float* lim = per_env_limits ? b + L.lim + lane * lim_row : smem + L.lim;
        float acc = 0.0f;
        for (int j = 0; j < nd; ++j) {
          const float x = s_img[lane * KD + j];
          float o = -fminf(x - lim[2 * j], 0.0f);
          o += fmaxf(x - lim[2 * j + 1], 0.0f);
          acc += o;
        }
        s_red[1 * T + lane] = acc;
      } else {
        float acc = 0.0f;
        for (int j = 0; j < nd; ++j) { const float a = s_img[lane * KD + nd + j]; acc += a * a; }
        s_red[3 * T + lane] = acc;
      }
    }
    __syncthreads();

    // ---- task reward -------------------------------------------------------------------------------
    if (env_lane) {
      if (!g1) {
        bf.reward[env] = 1.0f;  // humanoid_amp_env.py:128-129
      } else {
        const float r_term = p.s_term * (float)died;
        const float r_act = p.s_act * s_red[lane];
        const float r_lim = p.s_lim * s_red[T + lane];
        const float r_acc = p.s_acc * s_red[2 * T + lane];
        const float r_vel = p.s_vel * s_red[3 * T + lane];
        const float basic = (((r_term + r_act) + r_lim) + r_acc) + r_vel;
        float track = 0.0f, err = 0.0f;
        if (p.use_command) {
          // g1_amp_env.py:249-265: planar body-frame velocity error, exp reward with linear floor (:500-532)
          const float* cmd = b + L.cmd + lane * 2;
          const Vec3 vb = quat_rotate_inverse_ref(Quat{rq[0], rq[1], rq[2], rq[3]}, Vec3{rl[0], rl[1], rl[2]});
          const float dx = vb.x - cmd[0];
          const float dy = vb.y - cmd[1];
          err = sqrtf(dx * dx + dy * dy);
          const float e2 = err * err;
          const float lin = p.val_at_thr - p.slope * (e2 - p.thr);
          const float ex = p.w_track * expf(-e2 / p.sigma_sq);
          track = e2 > p.thr ? lin : ex;
        }
        const float total = basic + track;
        bf.reward[env] = total;
        if (bf.reward_terms) {
          float* t = bf.reward_terms + env;
          t[0 * N] = total; t[1 * N] = track; t[2 * N] = err; t[3 * N] = r_term;
          t[4 * N] = r_act; t[5 * N] = r_lim; t[6 * N] = r_acc; t[7 * N] = r_vel;
        }
      }
    }

    // ---- outputs: LDS image -> HBM ----------------------------------------------------------------------------
    {  // AMP buffer: the tile's rows are one contiguous 16-B aligned span
      const f4* img4 = reinterpret_cast<const f4*>(s_img);
      f4* dst4 = reinterpret_cast<f4*>(bf.amp_obs_buffer + tile_base * KD);
      for (int i = tid; i < T * KD / 4; i += kBlock) dst4[i] = img4[i];
    }
    if (fused) {
      // the same rows, scaled, as the discriminator's input.  A lane owns the column pair (c, c + 1) -- c even, so both
      // sit in one 32-column k-block -- and walks rows r0, r0 + step, ...; `scaled` / `blocks` are compile-time inside
      // the row loop (one instantiation per combination, picked once)
      uint32_t* const xs = reinterpret_cast<uint32_t*>(bf.disc_input) + tile_base * pitch;
      auto column_pair = [&](auto scaled_c, auto blocks_c, const int c, const int r0, const int step) {
        constexpr bool kScaled = decltype(scaled_c)::value, kBlocks = decltype(blocks_c)::value;
        env_f2 m = {0.0f, 0.0f}, d = {1.0f, 1.0f};
        if (kScaled) {
          m = *reinterpret_cast<const env_f2*>(s_mu + c);
          d = *reinterpret_cast<const env_f2*>(s_dn + c);
        }
        const float* src = s_img + r0 * KD + c;
        uint32_t* dst = xs + r0 * pitch + (kBlocks ? (c >> 5) * 32 + ((c & 31) >> 1) : c);
#pragma unroll 2
        for (int r = r0; r < T; r += step, src += step * KD, dst += step * pitch) {
          const env_f2 v = *reinterpret_cast<const env_f2*>(src);
          float x0 = v.x, x1 = v.y;
          if (kScaled) {  // same operations, in the same order, as disc.hip's scaler passes
            x0 = (x0 - m.x) / d.x;  // skrl RunningStandardScaler, exact fp32 divide
            x1 = (x1 - m.y) / d.y;
            x0 = fminf(fmaxf(x0, -clip), clip);
            x1 = fminf(fmaxf(x1, -clip), clip);
          }
          if (kBlocks) {
            // p0 = rn16(v), p1 = rn16(v - p0) (plane_pair) for both columns at once: the p0 halves are one 32-bit word
            // of the block's first 64 B, the p1 halves the word 64 B further
            const env_f2 y = {x0 * s_x, x1 * s_x};
            const env_h2 a = __builtin_convertvector(y, env_h2);
            const env_f2 af = __builtin_convertvector(a, env_f2);
            const env_f2 rem = {y.x - af.x, y.y - af.y};
            const env_h2 lo = __builtin_convertvector(rem, env_h2);
            dst[0] = __builtin_bit_cast(uint32_t, a);
            dst[16] = __builtin_bit_cast(uint32_t, lo);
          } else {
            uint2 o;
            o.x = __float_as_uint(x0);
            o.y = __float_as_uint(x1);
            *reinterpret_cast<uint2*>(dst) = o;
          }
        }
      };
      const int PR = KD >> 1;  // column pairs per row (K*D is even: checked on the host)
      auto walk = [&](auto scaled_c, auto blocks_c) {
        if (PR >= kBlock) {
          for (int c2 = tid; c2 < PR; c2 += kBlock) column_pair(scaled_c, blocks_c, 2 * c2, 0, 1);
        } else {
          const int G = kBlock / PR, g = row_of(tid, 1.0f / (float)PR);
          if (g < G) column_pair(scaled_c, blocks_c, 2 * (tid - g * PR), g, G);
        }
      };
      using yes = std::integral_constant<bool, true>;
      using no = std::integral_constant<bool, false>;
      if (scaled) { if (blocks) walk(yes{}, yes{}); else walk(yes{}, no{}); }
      else { if (blocks) walk(no{}, yes{}); else walk(no{}, no{}); }
    }
    {  // policy observation (g1_amp_env.py:195-242; humanoid_amp_env.py:126), P == Pcur (no actor history here)
      float* pol = bf.policy_obs + tile_base * P;
      const int img_off = (int)(s_img - smem), la_off = (int)(b + L.la - smem), cmd_off = (int)(b + L.cmd - smem);
      auto source = [&](const int c, int& src_pitch) -> const float* {  // LDS column c of the policy row (selects, no table)
        const bool in_img = !extra || c < Db, in_la = c < Db + nd;
        src_pitch = in_img ? KD : (in_la ? nd : 2);
        return smem + (in_img ? img_off + c : (in_la ? la_off + (c - Db) : cmd_off + (c - Db - nd)));
      };
      if ((P & 1) == 0) {
        auto column_pair = [&](const int c, const int r0, const int step) {
          int pa, pb;
          const float* a = source(c, pa);
          const float* bq = source(c + 1, pb);
          a += r0 * pa;
          bq += r0 * pb;
          float* dst = pol + (int64_t)r0 * P + c;
#pragma unroll 2
          for (int r = r0; r < T; r += step, a += step * pa, bq += step * pb, dst += step * P) {
            float2 o;
            o.x = *a;
            o.y = *bq;
            *reinterpret_cast<float2*>(dst) = o;
          }
        };
        const int PR = P >> 1;
        if (PR >= kBlock) {
          for (int c2 = tid; c2 < PR; c2 += kBlock) column_pair(2 * c2, 0, 1);
        } else {
          const int G = kBlock / PR, g = row_of(tid, 1.0f / (float)PR);
          if (g < G) column_pair(2 * (tid - g * PR), g, G);
        }
      } else {
        auto column = [&](const int c, const int r0, const int step) {
          int pa;
          const float* a = source(c, pa) + r0 * pa;
          float* dst = pol + (int64_t)r0 * P + c;
          for (int r = r0; r < T; r += step, a += step * pa, dst += step * P) *dst = *a;
        };
        if (P >= kBlock) {
          for (int c = tid; c < P; c += kBlock) column(c, 0, 1);
        } else {
          const int G = kBlock / P, g = row_of(tid, 1.0f / (float)P);
          if (g < G) column(tid - g * P, g, G);
        }
      }
    }
  }
}

