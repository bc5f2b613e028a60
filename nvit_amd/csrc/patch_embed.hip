// Dual patch embedding as ONE kernel (reference /root/reference/nvit/model.py:286-304 builds the two Conv2d
// patchifiers, :407-415 applies them and adds the position embeddings):
//   loc[m][:] = W_l . patch_l(m) + b_l + pos_l[t]      patch_l = the Pl x Pl pixels of token m           (K = ch*Pl*Pl)
//   glo[m][:] = W_g . patch_g(m) + b_g + pos_g[t]      patch_g = the Pg x Pg window centred on it, reflect padded
// The im2col matrix never exists in HBM.  A workgroup owns 256 tokens x 256 embedding channels of one side; per
// stage of 32 patch elements its threads gather the fp32 pixels straight from the image (float4 runs of a patch row,
// served by L2: neighbouring windows overlap), split them into bf16 hi + lo and write both halves of the MFMA operand
// tile into LDS, while the matching [hi32 | lo32] slice of the weight image arrives by LDS-DMA.  Three
// v_mfma_f32_16x16x32_bf16 products per fragment pair (hi*hi + lo*hi + hi*lo) give the fp32-accurate result the
// precision policy of the bf16 mode asks for (model.py::_EmbedFn); bias and position embedding are added in the
// epilogue, which writes whole 128-byte rows.  The workgroups that own channel tile 0 also store the bf16 hi image of
// the patches ([M, Kp], 1/3 of what the split im2col used to write) - the saved operand of the weight-gradient GEMM.
#include "gemm_common.h"

namespace {

constexpr int PE_TM = 256, PE_TN = 256;
constexpr int PE_A_BYTES = PE_TM * 128, PE_W_BYTES = PE_TN * 128, PE_SLOT = PE_A_BYTES + PE_W_BYTES;
constexpr int PE_LDS = 2 * PE_SLOT;  // 128 KiB ring; the epilogue scratch (8 x 2 KiB) reuses slot 0

typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

struct PeSide {
  const char* W;      // [C][2*Kp] bf16, per 32 patch elements: [hi32 | lo32]   (nvit_shadow_weights, perm 2)
  const float* bias;  // [C] or NULL
  const float* pos;   // [T][C]
  float* out;         // [M][C]
  bf16* out_lo;       // [M][C] bf16 twin of out (the operand of the next GEMM) or NULL
  bf16* a_hi;         // [tiles_m*256][Kp] or NULL
  int P, pad, K, Kp;  // patch edge, reflect pad, patch length ch*P*P and its padding to whole stages
};
struct PeArgs {
  const float* img;  // [B][ch][S][S]
  int ch, S, G, T, M, C, stride;
  int tiles_m, tiles_n, mgroups;
  PeSide side[2];  // 0 = global (4x the work of local: dealt first), 1 = local
};

__device__ __forceinline__ int pe_reflect(int i, int n) {
  i = i < 0 ? -i : i;
  return i >= n ? 2 * (n - 1) - i : i;
}

template <int N>
__device__ __forceinline__ void pe_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <bool VEC>
__global__ __launch_bounds__(512) void patch_embed_kernel(PeArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 2, wc = wid & 3;  // wave sub-tile: 128 tokens x 64 channels
  const int l15 = lane & 15, lg = lane >> 4;
  // Work item.  Workgroups go round-robin to the 8 XCDs: XCD x takes the token tiles = x (mod 8) and walks the channel
  // tiles of one token tile back to back, so the pixels a tile gathers are fetched into that XCD's L2 once.
  const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
  const int nt_i = q % g.tiles_n;
  int mg = q / g.tiles_n;
  const int sd = mg >= g.mgroups ? 1 : 0;
  if (sd) mg -= g.mgroups;
  const int mt = mg * 8 + xcd;
  if (mt >= g.tiles_m) return;
  const PeSide& sp = g.side[sd];
  const int m0 = mt * PE_TM, n0 = nt_i * PE_TN;
  const int nst = sp.Kp >> 5;
  const int P = sp.P, PP = P * P, K = sp.K, S = g.S;
  bf16* const a_hi = nt_i == 0 ? sp.a_hi : nullptr;

  // ---- producer: thread -> token rows prow and prow + 128, patch elements 8*o .. 8*o + 7 of every stage
  const int prow = tid >> 2, o = tid & 3;
  const float* ibase[2];
  int y0[2], x0[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    int m = m0 + prow + h * 128;
    m = m < g.M ? m : g.M - 1;
    const int b = m / g.T, t = m - b * g.T;
    const int ty = t / g.G, tx = t - ty * g.G;
    ibase[h] = g.img + (size_t)b * g.ch * S * S;
    y0[h] = ty * g.stride - sp.pad;
    x0[h] = tx * g.stride - sp.pad;
  }
  f32x4 v[2][2];
  // Branch-free on purpose (a conditional load into the same registers makes hipcc wait for the weight DMA in front
  // of every gather).  VEC (pad % 4 == 0, the usual geometry): a 4-pixel run starts at a multiple of 4, so it lies
  // entirely inside the row or entirely in the reflected border, where it is a contiguous run read backwards - one
  // unaligned 16-byte load plus a select either way.  Otherwise four scalar loads with per-pixel reflection.
  unsigned fix = 0;   // per run: bit (2h+e) = read backwards (reflected border), bit 4+e = past K (zero padding)
  auto gather = [&](int j) {
    fix = 0;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int k = j * 32 + o * 8 + e * 4;
      const int kc = k < K ? k : K - 4;
      const int c = kc / PP, rem = kc - c * PP, ph = rem / P, pw = rem - ph * P;
      fix |= k >= K ? 16u << e : 0u;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int y = pe_reflect(y0[h] + ph, S), x = x0[h] + pw;
        const float* row = ibase[h] + ((size_t)c * S + y) * S;
        if constexpr (VEC) {
          const bool rev = x < 0 || x >= S;
          const int xs = x < 0 ? -x - 3 : (x >= S ? 2 * (S - 1) - x - 3 : x);
          v[h][e] = *reinterpret_cast<const f32x4u*>(row + xs);
          fix |= rev ? 1u << (2 * h + e) : 0u;
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) v[h][e][i] = row[pe_reflect(x + i, S)];
        }
      }
    }
  };
  auto scatter = [&](int slot, int j) {
    uint4 keep[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int r = prow + h * 128;
      bf16x8 hi, lo;
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float x = (fix >> (2 * h + e)) & 1u ? v[h][e][3 - i] : v[h][e][i];
          x = (fix >> (4 + e)) & 1u ? 0.f : x;
          const bf16 hh = (bf16)x;
          hi[e * 4 + i] = hh;
          lo[e * 4 + i] = (bf16)(x - (float)hh);
        }
      char* arow = smem + slot * PE_SLOT + r * 128;
      keep[h] = __builtin_bit_cast(uint4, hi);
      *reinterpret_cast<uint4*>(arow + ((o ^ (r & 7)) << 4)) = keep[h];
      *reinterpret_cast<uint4*>(arow + (((4 + o) ^ (r & 7)) << 4)) = __builtin_bit_cast(uint4, lo);
    }
    // (both stores last and together, and rows past M exist in the padded a_hi buffer: exactly two store instructions
    //  per wave follow the DMA in the vmcnt queue, which is what the counted wait at the end of the stage relies on)
    if (a_hi) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
        *reinterpret_cast<uint4*>(a_hi + (size_t)(m0 + prow + h * 128) * sp.Kp + j * 32 + o * 8) = keep[h];
    }
  };

  // ---- weight slice by LDS-DMA: one wave-instruction = 8 channel rows x 128 B, chunks XOR-swizzled by the row
  const unsigned lds_base = (unsigned)(uintptr_t)((__attribute__((address_space(3))) void*)smem);
  const int srow = lane >> 3, gc = (lane & 7) ^ srow;
  const char* wp[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int row = n0 + (i * 8 + wid) * 8 + srow;
    row = row < g.C ? row : g.C - 1;
    wp[i] = sp.W + ((size_t)row * 2 * sp.Kp + gc * 8) * sizeof(bf16);
  }
  auto dma_w = [&](int j, int slot) {
    const unsigned bo = lds_base + (unsigned)(slot * PE_SLOT + PE_A_BYTES + wid * 1024);
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(wp[i] + (size_t)j * 128, bo + i * 8192);
  };

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  dma_w(0, 0);
  gather(0);
  scatter(0, 0);
  pe_wait_vmcnt<0>();
  __syncthreads();

  for (int s = 0; s < nst; ++s) {
    const int slot = s & 1;
    // (the last stage repeats itself into the idle slot instead of branching: a loop body without conditionals keeps
    //  hipcc's own vmcnt bookkeeping from draining the DMA queue at the loop head)
    const int nx = s + 1 < nst ? s + 1 : s;
    dma_w(nx, slot ^ 1);   // the other slot was released by the barrier that ended stage s - 1
    gather(nx);            // pixels of the next stage travel under this stage's MFMAs
    __builtin_amdgcn_sched_barrier(0);
    {
      const char* la = smem + slot * PE_SLOT;
      const char* lb = la + PE_A_BYTES;
      uint4 bh[4], bl[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = wc * 64 + j * 16 + l15;
        bh[j] = *reinterpret_cast<const uint4*>(lb + r * 128 + ((lg ^ (r & 7)) << 4));
        bl[j] = *reinterpret_cast<const uint4*>(lb + r * 128 + (((4 + lg) ^ (r & 7)) << 4));
      }
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        uint4 ah[4], al[4];
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
          const int r = wr * 128 + (half * 4 + ii) * 16 + l15;
          ah[ii] = *reinterpret_cast<const uint4*>(la + r * 128 + ((lg ^ (r & 7)) << 4));
          al[ii] = *reinterpret_cast<const uint4*>(la + r * 128 + (((4 + lg) ^ (r & 7)) << 4));
        }
#pragma unroll
        for (int ii = 0; ii < 4; ++ii)
#pragma unroll
          for (int j = 0; j < 4; ++j) Mma<bf16>::run(bh[j], ah[ii], acc[half * 4 + ii][j]);
#pragma unroll
        for (int ii = 0; ii < 4; ++ii)
#pragma unroll
          for (int j = 0; j < 4; ++j) Mma<bf16>::run(bh[j], al[ii], acc[half * 4 + ii][j]);
#pragma unroll
        for (int ii = 0; ii < 4; ++ii)
#pragma unroll
          for (int j = 0; j < 4; ++j) Mma<bf16>::run(bl[j], ah[ii], acc[half * 4 + ii][j]);
      }
    }
    scatter(slot ^ 1, nx);
    // the weight DMA of stage s+1 must have landed; the (younger) a_hi stores may stay in flight
    if (a_hi)
      pe_wait_vmcnt<2>();
    else
      pe_wait_vmcnt<0>();
    __syncthreads();
  }

  // ---- epilogue: + bias + pos[t], whole-row stores through the per-wave scratch (ring slot 0 is dead now)
  NtArgs e;
  e.C = sp.out;
  e.M = g.M;
  e.N = g.C;
  e.ldc = g.C;
  e.bias = sp.bias;
  e.colscale = nullptr;
  e.rowadd = sp.pos;
  e.rowadd_period = g.T;
  e.accumulate = 0;
  nt_store_tile_staged<8, float>(e, acc, m0 + wr * 128, n0 + wc * 64, lane, smem + wid * 2048);
  if (sp.out_lo) {
    e.C = sp.out_lo;
    nt_store_tile_staged<8, bf16>(e, acc, m0 + wr * 128, n0 + wc * 64, lane, smem + wid * 2048);
  }
}

}  // namespace

// Padded patch length of the operand images (a whole number of 32-element stages).
extern "C" int nvit_patch_embed_kp(int K) { return (K + 31) / 32 * 32; }

extern "C" int nvit_patch_embed_fwd(const float* img, const void* w_l, const float* b_l, const float* pos_l, float* out_l,
                                    void* lo_l, void* a_l, const void* w_g, const float* b_g, const float* pos_g,
                                    float* out_g, void* lo_g, void* a_g, int B, int ch, int S, int Pl, int Pg, int C,
                                    void* stream) {
  NVIT_REQUIRE(Pl % 4 == 0 && Pg % 4 == 0 && Pl > 0 && S % Pl == 0 && Pg >= Pl && (Pg - Pl) % 2 == 0,
               "patch_embed: unsupported patch geometry S=%d Pl=%d Pg=%d", S, Pl, Pg);
  NVIT_REQUIRE((Pg - Pl) / 2 < S && S >= 4, "patch_embed: reflect pad must be smaller than the image");
  NVIT_REQUIRE(C % 4 == 0 && B > 0 && ch > 0 && (C % 8 == 0 || (!lo_l && !lo_g)), "patch_embed: bad C=%d", C);
  hipStream_t s = (hipStream_t)stream;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)patch_embed_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, PE_LDS);
    if (e == hipSuccess)
      e = hipFuncSetAttribute((const void*)patch_embed_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, PE_LDS);
    if (e != hipSuccess) NVIT_FAIL((int)e, "patch_embed: cannot raise LDS limit: %s", hipGetErrorString(e));
    attr_set = true;
  }
  PeArgs g;
  g.img = img;
  g.ch = ch;
  g.S = S;
  g.G = S / Pl;
  g.T = g.G * g.G;
  g.M = B * g.T;
  g.C = C;
  g.stride = Pl;
  g.tiles_m = cdiv(g.M, PE_TM);
  g.tiles_n = cdiv(C, PE_TN);
  g.mgroups = cdiv(g.tiles_m, 8);
  const int Kl = ch * Pl * Pl, Kg = ch * Pg * Pg;
  g.side[0] = PeSide{(const char*)w_g, b_g, pos_g, out_g, (bf16*)lo_g, (bf16*)a_g, Pg, (Pg - Pl) / 2, Kg, nvit_patch_embed_kp(Kg)};
  g.side[1] = PeSide{(const char*)w_l, b_l, pos_l, out_l, (bf16*)lo_l, (bf16*)a_l, Pl, 0, Kl, nvit_patch_embed_kp(Kl)};
  const int grid = 2 * g.mgroups * 8 * g.tiles_n;
  ProfScope ps(NVIT_KID_PATCHIFY, 6.0 * g.M * (double)C * (Kl + Kg),
               (double)B * ch * S * S * 4.0 + 2.0 * g.M * (double)C * (lo_l ? 6.0 : 4.0), s);
  if (((Pg - Pl) / 2) % 4 == 0 && S % 4 == 0)
    hipLaunchKernelGGL(patch_embed_kernel<true>, dim3(grid), dim3(512), PE_LDS, s, g);
  else
    hipLaunchKernelGGL(patch_embed_kernel<false>, dim3(grid), dim3(512), PE_LDS, s, g);
  NVIT_CHECK_LAUNCH("patch_embed");
  return NVIT_OK;
}
