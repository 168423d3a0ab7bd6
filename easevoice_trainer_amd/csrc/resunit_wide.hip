// resunit_wide: one HiFi-GAN ResBlock1 step  y = x + c2(lrelu(c1(lrelu(x))))  (src/easevoice/module/modules.py:299-308 of
// the reference) -- forward, and the data half of its backward -- as ONE launch for the WIDE vocoder stages (C = 64 at
// 5120 samples per item, C = 128 at 2560), gfx950 bf16.  Unfused, a step is a leaky-relu launch and two convolution
// launches each way, every one of them a 20-40 us kernel on a 40960 / 81920-position problem that fills 256 CUs with
// 320 blocks; here the activated intermediate never leaves LDS and a launch has exactly as many blocks as the chip
// has CUs (160-position tiles: 256 blocks at C = 128, 512 = two per CU at C = 64).
//
//   forward :  mid_a = lrelu(c1(xa) + b1) on the tile + c2's halo,  y = c2(mid_a) + b2 + x          (xa = lrelu(x))
//   backward:  dmid  = (c2^T dy) * lrelu'(mid_a) on the tile + c1's halo,  dx = (c1^T dmid) * lrelu'(xa) + dy
// Both are "first convolution on tile + halo into LDS, second convolution on the tile": one kernel, two epilogue sets.
// The weight gradients stay their own launches (wgrad_halo / wgrad_deep on xa / mid_a / dy / dmid, side stream).
//
// One block = one position tile x ALL channels.  The four waves form a WM x WN grid: WM waves split the output
// channels, WN waves split the positions.
//   * B operand (activations): the tile's rows sit in LDS once (2C bytes per row, 16-byte slots XOR-swizzled by the
//     row so that the 16-byte reads of lanes n + tap shift are conflict-free at every shift); all waves read them.
//   * A operand (weights): a wave needs only ITS output-channel rows of the prepared image, and every (row, K step)
//     fragment exactly once -- so the fragments go from global memory (L2-resident: every block reads the same image)
//     straight into registers, two K steps ahead of their MFMAs; no LDS traffic, no block barrier in the K loop.
//     A fragment feeds NT position tiles (9-14 MFMAs), which keeps the per-CU vector-memory path at 20-30 B/clk.
//   * the K loop runs over (chunk of 32 input channels, tap) = the order of the prepared images; B fragments of the next
//     K step are requested before the MFMAs of this one (software pipelining by hand, see resunit_common.h).
#include "resunit_common.h"
#include "../../include/evt.h"
#include <cstdlib>

namespace {

using namespace evt_ru;

struct WUP {
  const h16_t* in;      // forward: x (leaky-relu applied on load); backward: dy
  const h16_t* g1;      // backward: mid_a (gate of the first convolution's output); forward: null
  const h16_t* g2;      // backward: xa (gate of the second convolution's output); forward: null
  const h16_t* wA; const h16_t* wB;   // images of the first / second convolution (fwd: REG1, REG2; bwd: ALT2, ALT1)
  const float* bA; const float* bB;     // forward biases; backward null
  h16_t* in_act;        // forward: xa = lrelu(x) out (own rows) or null
  h16_t* outA;          // forward: mid_a, backward: dmid (own rows) or null
  h16_t* outB;          // forward: y, backward: dx
  int nseq, L, k, dilA, dilB;
  float slope, in_scale;
  int tps;               // tiles per sequence
  int xrows;             // staged input rows = P + 2 (hA + hB)
};

template <int C> __device__ __forceinline__ int wslot(int row, int slot) {
  // 16-byte slot `slot` of row `row` -> slot index inside the row: rows per 256-byte bank window = 512 / (2C)
  if constexpr (C == 64) return slot ^ (((row >> 1) & 3) << 1);
  else return slot ^ ((row & 7) << 1);
}

// C channels; WM x WN waves; NT1 / NT2: position tiles of 16 per wave of the first / second convolution
//
// Every global round trip of a block except the first is hidden: the tile's rows, the rows the second epilogue needs
// (forward: raw x for the residual; backward: xa for the gate) and -- backward -- the gate rows of the first epilogue
// (mid_a, parked in the intermediate tile's own LDS rows: each lane reads its 8 bytes right before it overwrites them
// with dmid) are all staged in the one pass at the top; the first two weight fragments of a convolution are requested
// before the barrier in front of it; biases at the very top.
template <int C, int WM, int WN, int NT1, int NT2, bool BWD, int R>
__global__ __launch_bounds__(256, 2) void resunit_wide(WUP p) {
  constexpr int PITCH = C * 2;
  constexpr int SPR = C / 8;                 // 16-byte slots per row
  constexpr int MTW = C / 16 / WM;           // output-channel tiles per wave
  constexpr int P = 16 * NT2 * WN;           // own positions per block
  constexpr int MROWS = 16 * NT1 * WN;       // rows of the intermediate tile
  constexpr int NCH = C / 32;                // chunks of 32 input channels
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  unsigned char* xs = smem;
  unsigned char* ms = smem + ((p.xrows + 7) & ~7) * PITCH;
  unsigned char* rs = ms + MROWS * PITCH;    // P rows: forward raw x, backward xa (own positions)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, g = lane >> 4;
  const int wm = wave % WM, wn = wave / WM;
  const int tile = blockIdx.x;
  const int seq = tile / p.tps;
  const int q0 = (tile - seq * p.tps) * P;
  const long sbase = (long)seq * p.L * C;
  const int H = (p.k - 1) / 2, hA = p.dilA * H, hB = p.dilB * H;
  const int ktot = NCH * p.k * 32;           // K elements per image row
  const int nks = NCH * p.k;                 // K steps of 32 (always even: NCH is)
  const int last = nks - 1;

  float biasA[MTW][4], biasB[MTW][4];
#pragma unroll
  for (int i = 0; i < MTW; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      biasA[i][r] = (!BWD && p.bA) ? p.bA[(wm * MTW + i) * 16 + g * 4 + r] : 0.f;
      biasB[i][r] = (!BWD && p.bB) ? p.bB[(wm * MTW + i) * 16 + g * 4 + r] : 0.f;
    }

  // one cooperative pass: `rows` rows of `src` starting at position pos0 -> region `dst` (zero outside the sequence)
  auto stage = [&](const h16_t* src, unsigned char* dst, const int rows, const int pos0, auto xform) {
    const int npieces = rows * SPR;
    for (int base = 0; base < npieces; base += 256 * 8) {
      uint4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = base + u * 256 + tid;
        const int r = idx / SPR, sl = idx - r * SPR;
        const int pos = pos0 + r;
        v[u] = (idx < npieces && pos >= 0 && pos < p.L) ? *reinterpret_cast<const uint4*>(src + sbase + (long)pos * C + sl * 8)
                                                        : make_uint4(0, 0, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = base + u * 256 + tid;
        if (idx >= npieces) continue;
        const int r = idx / SPR, sl = idx - r * SPR;
        xform(v[u], r, sl, dst + r * PITCH + wslot<C>(r, sl) * 16);
      }
    }
  };
  if constexpr (!BWD) {
    stage(p.in, xs, p.xrows, q0 - hA - hB, [&](const uint4 raw, const int r, const int sl, unsigned char* d) {
      const uint4 a = lrelu8(raw, p.slope);
      *reinterpret_cast<uint4*>(d) = a;
      const int o = r - hA - hB;
      if (o >= 0 && o < P) {
        *reinterpret_cast<uint4*>(rs + o * PITCH + wslot<C>(o, sl) * 16) = raw;
        if (p.in_act && q0 + o < p.L) *reinterpret_cast<uint4*>(p.in_act + sbase + (long)(q0 + o) * C + sl * 8) = a;
      }
    });
  } else {
    stage(p.in, xs, p.xrows, q0 - hA - hB, [&](const uint4 raw, const int, const int, unsigned char* d) {
      *reinterpret_cast<uint4*>(d) = p.in_scale != 1.f ? scale8(raw, p.in_scale) : raw;
    });
    stage(p.g1, ms, MROWS, q0 - hB, [&](const uint4 raw, const int, const int, unsigned char* d) { *reinterpret_cast<uint4*>(d) = raw; });
    stage(p.g2, rs, P, q0, [&](const uint4 raw, const int, const int, unsigned char* d) { *reinterpret_cast<uint4*>(d) = raw; });
  }

  // weight fragments (global, L2-resident): a ring of R sets, requested R - 2 K steps ahead (4 waves x (R - 2) x MTW KB in
  // flight per block; deeper rings measured alike, see evt_resunit_wide_fwd); K steps behind the last one re-load the last
  // fragment (clamped index) so that no load sits behind a branch
  u32x4 fa[R][MTW];
  const h16_t* wrow[MTW];
  auto set_w = [&](const h16_t* w) {
#pragma unroll
    for (int i = 0; i < MTW; ++i) wrow[i] = w + (long)((wm * MTW + i) * 16 + n) * ktot + g * 8;
  };
  auto issue_a = [&](const int ks, const int s) {
    const int kc = ks < last ? ks : last;
#pragma unroll
    for (int i = 0; i < MTW; ++i) fa[s][i] = *reinterpret_cast<const u32x4*>(wrow[i] + kc * 32);
  };
  auto prologue_a = [&](const h16_t* w) {
    set_w(w);
#pragma unroll
    for (int i = 0; i < R - 2; ++i) issue_a(i, i);
  };
  // one convolution: acc[i][j] over the K steps; rows region `rows`, first row of this wave's tile j = rbase + 16 j.
  // Straight-line K loop (no branch around a load in the main part: behind a control-flow merge the compiler's wait
  // counters fall back to "wait for everything", which would drain the weight prefetch every step); R K steps per trip so
  // the weight ring rotates without register copies.  Rows (LDS): ONE fragment set; tile j's fragment of the next K step
  // is requested right after tile j's MFMAs of this step were issued, i.e. a whole K step before it is needed.
  // Precondition: prologue_a(w) already done (before the barrier in front of the convolution).
  auto conv = [&](auto& acc, auto NT_c, const unsigned char* rows, const int rbase, const int dil) {
    constexpr int NT = decltype(NT_c)::value;
    u32x4 fb[NT];
    int bch = 0, btap = 0;                // (chunk, tap) of the K step whose row fragments are requested next
    auto b_base = [&]() {
      const int row = rbase + n + btap * dil;
      const unsigned char* base = rows + row * PITCH + wslot<C>(row, bch * 4 + g) * 16;
      if (bch * p.k + btap < last) { ++btap; if (btap == p.k) { btap = 0; ++bch; } }     // clamps at the last K step
      return base;
    };
    auto step = [&](const int sa) {
#pragma unroll
      for (int i = 0; i < MTW; ++i) tie(fa[sa][i]);
      const unsigned char* nb = b_base();
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        tie(fb[j]);
#pragma unroll
        for (int i = 0; i < MTW; ++i)
          acc[i][j] = EVT_MFMA_16x16x32(as_h8(fa[sa][i]), as_h8(fb[j]), acc[i][j], 0, 0, 0);
        fb[j] = *reinterpret_cast<const u32x4*>(nb + j * 16 * PITCH);
      }
    };
    {
      const unsigned char* b0 = b_base();
#pragma unroll
      for (int j = 0; j < NT; ++j) fb[j] = *reinterpret_cast<const u32x4*>(b0 + j * 16 * PITCH);
    }
    int ks = 0;
    for (; ks + R <= nks; ks += R) {
#pragma unroll
      for (int i = 0; i < R; ++i) {
        issue_a(ks + i + R - 2, (i + R - 2) % R);
        step(i);
      }
    }
    const int rest = nks - ks;            // < R K steps left; their weights are (being) fetched into sets 0 .. rest - 1
#pragma unroll
    for (int i = 0; i < R; ++i)
      if (i < rest) step(i);
  };
  auto unpack4 = [](const u32x2 v, float (&o)[4]) {
    o[0] = h2f_lo(v[0]); o[1] = h2f_hi(v[0]);
    o[2] = h2f_lo(v[1]); o[3] = h2f_hi(v[1]);
  };

  prologue_a(p.wA);
  __syncthreads();

  // ---- first convolution: rows m = wn * 16 NT1 + 16 j + n of the intermediate tile (position q0 - hB + m) ----
  {
    f32x4 acc[MTW][NT1];
#pragma unroll
    for (int i = 0; i < MTW; ++i)
#pragma unroll
      for (int j = 0; j < NT1; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int m0 = wn * 16 * NT1;
    conv(acc, std::integral_constant<int, NT1>{}, xs, m0, p.dilA);
    // the second convolution's first weights travel under this epilogue
    prologue_a(p.wB);
#pragma unroll
    for (int j = 0; j < NT1; ++j) {
      const int m = m0 + j * 16 + n;
      const int pos = q0 - hB + m;
      const bool inside = pos >= 0 && pos < p.L && m < P + 2 * hB;
#pragma unroll
      for (int i = 0; i < MTW; ++i) {
        const int c = (wm * MTW + i) * 16 + g * 4;
        unsigned char* mp = ms + m * PITCH + wslot<C>(m, c >> 3) * 16 + (c & 7) * 2;
        h16_t o4[4];
        if constexpr (BWD) {
          float gg[4];
          unpack4(*reinterpret_cast<const u32x2*>(mp), gg);          // mid_a, staged where dmid goes
#pragma unroll
          for (int r = 0; r < 4; ++r) o4[r] = f2h(inside ? acc[i][j][r] * (gg[r] > 0.f ? 1.f : p.slope) : 0.f);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float v = acc[i][j][r] + biasA[i][r];
            v = v > 0.f ? v : v * p.slope;
            o4[r] = f2h(inside ? v : 0.f);
          }
        }
        *reinterpret_cast<uint2*>(mp) = *reinterpret_cast<uint2*>(o4);
        if (p.outA && inside && m >= hB && m < hB + P)
          *reinterpret_cast<uint2*>(p.outA + sbase + (long)pos * C + c) = *reinterpret_cast<uint2*>(o4);
      }
    }
  }
  __syncthreads();

  // ---- second convolution: own positions o = wn * 16 NT2 + 16 j + n; intermediate row of (o, tap) = o + tap * dilB ----
  {
    f32x4 acc[MTW][NT2];
#pragma unroll
    for (int i = 0; i < MTW; ++i)
#pragma unroll
      for (int j = 0; j < NT2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int o0 = wn * 16 * NT2;
    conv(acc, std::integral_constant<int, NT2>{}, ms, o0, p.dilB);
#pragma unroll
    for (int j = 0; j < NT2; ++j) {
      const int o = o0 + j * 16 + n;
      const int q = q0 + o;
      if (q >= p.L) continue;
#pragma unroll
      for (int i = 0; i < MTW; ++i) {
        const int c = (wm * MTW + i) * 16 + g * 4;
        float rr[4];
        h16_t o4[4];
        if constexpr (BWD) {
          float gg[4];
          const int r0 = hA + hB + o;                                // the (scaled) dy row of this position
          unpack4(*reinterpret_cast<const u32x2*>(xs + r0 * PITCH + wslot<C>(r0, c >> 3) * 16 + (c & 7) * 2), rr);
          unpack4(*reinterpret_cast<const u32x2*>(rs + o * PITCH + wslot<C>(o, c >> 3) * 16 + (c & 7) * 2), gg);
#pragma unroll
          for (int r = 0; r < 4; ++r) o4[r] = f2h(acc[i][j][r] * (gg[r] > 0.f ? 1.f : p.slope) + rr[r]);
        } else {
          unpack4(*reinterpret_cast<const u32x2*>(rs + o * PITCH + wslot<C>(o, c >> 3) * 16 + (c & 7) * 2), rr);
#pragma unroll
          for (int r = 0; r < 4; ++r) o4[r] = f2h(acc[i][j][r] + biasB[i][r] + rr[r]);
        }
        *reinterpret_cast<uint2*>(p.outB + sbase + (long)q * C + c) = *reinterpret_cast<uint2*>(o4);
      }
    }
  }
}

template <int C, int WM, int WN, int NT1, int NT2, bool BWD, int R = 4>
int launch(WUP p, hipStream_t st) {
  constexpr int P = 16 * NT2 * WN, MROWS = 16 * NT1 * WN;
  const int H = (p.k - 1) / 2, hA = p.dilA * H, hB = p.dilB * H;
  if (P + 2 * hB > MROWS) return EVT_ENOTSUP;
  p.xrows = P + 2 * (hA + hB);
  // the first convolution of the last tile row reads input rows up to MROWS - 1 + (k - 1) dilA
  const int xneed = MROWS + (p.k - 1) * p.dilA;
  if (xneed > p.xrows) p.xrows = xneed;
  p.tps = (p.L + P - 1) / P;
  const size_t lds = (size_t)(((p.xrows + 7) & ~7) + MROWS + P) * C * 2;
  if (lds > 160 * 1024) return EVT_ENOTSUP;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&resunit_wide<C, WM, WN, NT1, NT2, BWD, R>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return EVT_ELAUNCH;
    attr = true;
  }
  evt_set_last_tag("resunit_wide_%s<bf16, %d, %dx%d, nt %d-%d>", BWD ? "bwd" : "fwd", C, WM, WN, NT1, NT2);
  hipLaunchKernelGGL((resunit_wide<C, WM, WN, NT1, NT2, BWD, R>), dim3(p.nseq * p.tps), dim3(256), lds, st, p);
  return evt_check_launch();
}

bool wide_ok(const evt_resunit_params* a) {
  if (!a || a->dtype != EVT_DT_HALF) return false;
  if (a->C != 64 && a->C != 128) return false;
  if (a->k != 3 && a->k != 7 && a->k != 11) return false;
  if (a->dil < 1 || a->dil > 5 || a->nseq <= 0 || a->L < 64) return false;
  return true;
}

}  // namespace

extern "C" {

int32_t evt_resunit_wide_supported(const evt_resunit_params* a) {
  static const bool off = getenv("EVT_NO_RESUNIT_WIDE") != nullptr;   // A/B switch for measurements
  return (!off && wide_ok(a)) ? 1 : 0;
}

int evt_resunit_wide_fwd(const evt_resunit_params* a, const void* x, const void* w1_reg, const void* w2_reg, const float* b1,
                         const float* b2, void* xa, void* mid_a, void* y, void* stream) {
  if (!evt_resunit_wide_supported(a)) return EVT_ENOTSUP;
  if (!x || !w1_reg || !w2_reg || !y) return EVT_EINVAL;
  WUP p{};
  p.in = (const h16_t*)x; p.wA = (const h16_t*)w1_reg; p.wB = (const h16_t*)w2_reg; p.bA = b1; p.bB = b2;
  p.in_act = (h16_t*)xa; p.outA = (h16_t*)mid_a; p.outB = (h16_t*)y;
  p.nseq = a->nseq; p.L = a->L; p.k = a->k; p.dilA = a->dil; p.dilB = 1; p.slope = a->slope; p.in_scale = 1.f;
  hipStream_t st = (hipStream_t)stream;
  // Tile shape, measured on the B = 16 shapes (us per launch, k = 3 / 7 / 11; tools/bench_resunit.py --wide-fwd):
  //   C = 128:  4x1 waves, 160 positions (256 blocks)  31 / 35 / 44   |  4x1, 80 positions (512 blocks)  32 / 42 / 60
  //             2x2 waves, 160 positions               32 / 48 / 64   |  2x2, 96 positions               42 / 72 / 103
  //   C = 64:   4x1 waves, 160 positions (512 blocks)  25 / 24 / 32   |  2x2, 160 positions              24 / 28 / 36
  //             2x2 waves, 256 positions (320 blocks)  26 / 33 / 41   |  4x1, 80 positions               26 / 28 / 36
  // i.e. what counts is weight bytes per MFMA: four waves along the channels (a wave fetches only its own rows) and as
  // many position tiles per wave as the accumulators allow.  The first convolution covers P + 2 * 5 rows: 176 >= 170.
  // ring depth R (weight fragment sets, requested R - 2 K steps ahead): 4, 5, 6, 8, 12 measured alike (C = 128, k = 11:
  // 40.9 / 41.1 / 41.3 / 41.6 / 41.3 us) -- the launch is not waiting for its weight stream.  Counters (rocprofv3 --pmc,
  // profiles/r04_resunit_pmc.txt): MFMA busy 16-24 %, LDS array busy 15-27 % (a tenth of that bank conflicts), 0.7-1.1
  // waves per SIMD: neither pipe is near its rate; one block of four waves per CU means every wave waits out its own
  // fragment-read -> MFMA chain with nothing else to run (t = 19 us + 2 x MFMA time over k = 3 / 7 / 11).
  if (a->C == 128) return launch<128, 4, 1, 11, 10, false, 4>(p, st);
  return launch<64, 4, 1, 11, 10, false, 4>(p, st);
}

int evt_resunit_wide_bwd_data(const evt_resunit_params* a, const void* dy, float dy_scale, const void* xa, const void* mid_a,
                              const void* w1_alt, const void* w2_alt, void* dmid, void* dx, void* stream) {
  if (!evt_resunit_wide_supported(a)) return EVT_ENOTSUP;
  if (!dy || !xa || !mid_a || !w1_alt || !w2_alt || !dx) return EVT_EINVAL;
  WUP p{};
  p.in = (const h16_t*)dy; p.g1 = (const h16_t*)mid_a; p.g2 = (const h16_t*)xa;
  p.wA = (const h16_t*)w2_alt; p.wB = (const h16_t*)w1_alt;
  p.outA = (h16_t*)dmid; p.outB = (h16_t*)dx;
  p.nseq = a->nseq; p.L = a->L; p.k = a->k; p.dilA = 1; p.dilB = a->dil; p.slope = a->slope; p.in_scale = dy_scale;
  hipStream_t st = (hipStream_t)stream;
  const int need = 2 * a->dil * ((a->k - 1) / 2);       // rows of the intermediate tile beyond the own positions
  if (a->C == 128) {
    if (160 + need <= 176) return launch<128, 4, 1, 11, 10, true>(p, st);
    if (160 + need <= 192) return launch<128, 4, 1, 12, 10, true>(p, st);
    return launch<128, 4, 1, 14, 10, true>(p, st);
  }
  if (160 + need <= 176) return launch<64, 4, 1, 11, 10, true>(p, st);
  if (160 + need <= 192) return launch<64, 4, 1, 12, 10, true>(p, st);
  return launch<64, 4, 1, 14, 10, true>(p, st);
}

}  // extern "C"
