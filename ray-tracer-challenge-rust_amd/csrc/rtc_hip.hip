// rtc_hip.hip — gfx950 megakernel + the C ABI declared in include/rtc_hip.h.
//
// Per pixel the kernel runs
//   ray_for_pixel (camera.rs:48-65) -> World::intersect over the leaves (world.rs:43-54, shape.rs:248-463)
//   -> Intersection::hit / prepare_computations (intersection.rs:17-83) -> shade_hit (world.rs:56-78) with
//   Material::lighting, the is_shadowed shadow ray, and reflected_color / refracted_color folded into a loop
//   with a small per-pixel ray stack (no recursion on the device).
// Two sweeps implement World::intersect:
//   sweep<>        brute force, the reference's algorithm: every ray x every leaf in f64, triangle runs staged
//                  through LDS in tiles shared by the four waves of a 256-thread workgroup (RTC_FLAG_EXACT_SWEEP);
//   sweep_accel<>  default: spatial clusters; tile, cluster and triangle boxes tested one per LANE against the
//                  wave's ray bundle, per-ray f32 box and triangle filters on what survives (records broadcast
//                  from registers), and the same exact f64 test on the filters' survivors; one wave per
//                  workgroup, no barriers.  Bit-identical results.
// The per-ray matrix inverse and Bounds::new that the reference recomputes are flatten-time constants here.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared (see build.py / __graft_entry__.py).
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <array>
#include <vector>

#include "../../include/rtc_hip.h"
#include "rtc_device.hpp"

#pragma clang fp contract(off)

#ifndef RTC_TRI_PAD
#define RTC_TRI_PAD 1e-6f  // padding of the per-triangle boxes, relative to the run's extent (bundle_tri); tests build a
                          // broken variant with a negative value to prove that they notice a lost hit
#endif
#ifndef RTC_BG_EXIT
#define RTC_BG_EXIT 1  // wave-level early out for waves whose primary rays miss every top-level group box
#endif
#ifndef RTC_SHADOW_EAGER
#define RTC_SHADOW_EAGER 2  // shadow sweeps run the queued exact tests after every tile (1) / cluster group (2) so that
                            // occluded rays leave early and a fully occluded wave stops; 0 = only when a queue is full
#endif
#ifndef RTC_BLOCK_FILTER
#define RTC_BLOCK_FILTER 16  // per-ray test of 8-cluster blocks when at least this many clusters of a tile survive the
                             // bundle test (0 = never)
#endif
#ifndef RTC_FRONT_TO_BACK
#define RTC_FRONT_TO_BACK 1  // visit tiles / clusters from the side the wave's rays come from
#endif
#ifndef RTC_BRANCH_FREE_ROUNDS
#define RTC_BRANCH_FREE_ROUNDS 7
#endif
#ifndef RTC_PARK_REFRACTED
#define RTC_PARK_REFRACTED 1  // of two children the reflected one continues in the lane, the refracted one is parked: C5's parked
                              // rays then fit the 64 LDS entries of a wave but for a handful (HBM writes 215 -> 103 MB per frame,
                              // the framebuffer is 99.5), at the same frame time once the refill threshold is 1 (profiles/r02/v28_*..v30_*)
#endif
#ifndef RTC_XCD_STRIP
#define RTC_XCD_STRIP 16  // tiles per strip dealt to one XCD (0: plain 2-D grid, consecutive tiles on different XCDs).  16: C4 -7 %,
                          // C5 -4 %, C3 -2 % against 0; 4 / 8 gain half of that, 32 / 64 lose balance (profiles/r02/v35_*)
#endif
#ifndef RTC_BUSY_FIRST
#define RTC_BUSY_FIRST 1  // C3 -5 %, C4 -5 %, C5 -7 %, hexagon -6 % (profiles/r02/v39_*)
#endif
#ifndef RTC_ROW_COST_ORDER
#define RTC_ROW_COST_ORDER 1  // the same view upside down: 0.145 -> 0.106 ms; BASELINE views within 1 % (profiles/r02/v44_*)
#endif
#ifndef RTC_STRIP_ORDER
#define RTC_STRIP_ORDER 1  // scenes with triangle runs: the launch takes its strips of RTC_XCD_STRIP tiles in descending order of an estimate
                           // of their longest wave's life (triangles per pixel of the mesh blocks that project onto a tile), from a
                           // table in device memory computed per (camera, rows) on the host; supersedes RTC_BUSY_FIRST /
                           // RTC_ROW_COST_ORDER for those scenes (profiles/r03/v3_*)
#endif
#ifndef RTC_SHADOW_HOME
#define RTC_SHADOW_HOME 0  // opaque kernels with triangle runs: six more LDS slots per wave (12.6 instead of 9.6 KiB) for the shadow ray, so
                           // that shadow sweeps too keep the object-space ray in registers (sweep_accel: HOME).  The C3 frame, bounded by
                           // its longest wave, gains 1 %; frames bounded by throughput lose 4-6 % (C3zoom, C3 at 4K, C4): profiles/r03/v5_*
#endif
#ifndef RTC_REACH_CULL
#define RTC_REACH_CULL 1  // opaque scenes: once every ray of the wave has a hit, mesh tiles, cluster groups and triangles that lie beyond
                          // the farthest of those hits along the bundle's axis are skipped at wave level (before their boxes are fetched)
#endif
#ifndef RTC_CLEAR_EMPTY
#define RTC_CLEAR_EMPTY 1  // strip_table: tiles onto which no leaf's bounds project (by a conservative projection on the host) are not
                           // rendered: one wave per sixteen of them stores BLACK (no intersection is possible: world.rs:93)
#endif
#ifndef RTC_TRI_COVER
#define RTC_TRI_COVER 1  // strip_table: tiles are declared empty triangle by triangle, not cluster box by cluster box (RtcScene::tri_world)
#endif
constexpr uint32_t TRI_WORLD_MAX = 1u << 16;  // (72 B of host memory and three projections per new view per triangle; beyond it the cluster boxes decide, as before)
#ifndef RTC_SPLIT4_ABOVE
#define RTC_SPLIT4_ABOVE 1.1  // ... and longer than this share as four quarter tiles
#endif
#ifndef RTC_SPLIT_ABOVE
#define RTC_SPLIT_ABOVE 1e9  // strip_table: tiles estimated to live longer than this share of the frame are rendered as two half tiles
                             // (never: quarters alone are better, profiles/r03/v3_* item 10)
#endif
#ifndef RTC_TILE_PREFETCH
#define RTC_TILE_PREFETCH 0
#endif
#ifndef RTC_BUNDLE_SGPR
#define RTC_BUNDLE_SGPR 0
#endif
#ifndef RTC_RADIANCE_EAGER
#define RTC_RADIANCE_EAGER 1  // opaque scenes: radiance sweeps also run the queued exact tests after every cluster group and
                              // then ignore everything beyond the nearest hit found so far
#endif
#ifndef RTC_EXACT_MIN_WAVES
#define RTC_EXACT_MIN_WAVES 3  // brute-force kernels (256-thread workgroups)
#endif
#ifndef RTC_PRIM_MIN_WAVES
#define RTC_PRIM_MIN_WAVES 4  // kernels of primitive-only scenes (no triangle machinery): at most 128 VGPRs
#endif
#ifndef RTC_BLK_LDC
#define RTC_BLK_LDC 1
#endif
#ifndef RTC_POOL_BUDGET_MB
#define RTC_POOL_BUDGET_MB 4096ull
#endif
#ifndef RTC_SPLIT_REFR
#define RTC_SPLIT_REFR 1  // transparent scenes with triangle runs: forward radiance sweep + backward sweep on demand (see LeafAccFwd)
#endif
#ifndef RTC_SKIP_UNUSED_SHADOW
#define RTC_SKIP_UNUSED_SHADOW 1
#endif
#ifndef RTC_REFR_EAGER
#define RTC_REFR_EAGER 0  // scenes with a transparent material: radiance sweeps also run the queued exact tests after
                          // every cluster group and narrow the parameter range on both sides (nearest hit in front, the
                          // two nearest container candidates behind)
#endif
#ifndef RTC_REFILL_BELOW
#define RTC_REFILL_BELOW 1   // scenes with a transparent material: lanes without work refill from the wave's pool of parked
                             // rays once fewer than this many lanes are busy (1: when none is; 64: whenever one is free).
                             // With the reflected child parked, 1..48 timed the same on C5 and 64 was 3 % slower
                             // (profiles/r02/v2_*); with the refracted child parked (RTC_PARK_REFRACTED) 1 is the fastest: C5
                             // 2.70 ms against 2.74 (16), 2.79 (32), 2.90 (64), profiles/r02/v29_*, v30_*
#endif
#ifndef RTC_REGION_TIMERS
#define RTC_REGION_TIMERS 0  // diagnostics build (tools/build_variants.sh): counting launches also accumulate, per wave, the
                             // shader-clock cycles spent in each part of the kernel and loop-round counts (rtc_debug_counters)
#endif
#ifndef RTC_WAVE_TIMELINE
#define RTC_WAVE_TIMELINE 0  // diagnostics build (tools/wave_timeline.py): every workgroup of a render launch records when it started and
                             // ended (s_memrealtime, 100 MHz), its XCD and its tile: who is resident when, and what the frame's tail is
#endif
#ifndef RTC_MIN_WAVES
#define RTC_MIN_WAVES 3  // __launch_bounds__ second argument: waves per SIMD the register allocator must allow
                         // (168 VGPRs; measured on MI355X: 3 beats 1/2 by ~20% and 4 spills, profiles/r01/v23_*)
#endif
#ifndef RTC_OPAQUE_MIN_WAVES
#define RTC_OPAQUE_MIN_WAVES 3  // mesh kernels of opaque scenes.  With the ray bundle and (during the shadow sweep) the normal in
                                // LDS and 1/d recomputed where used they need 160 VGPRs; at 4 waves (128 VGPRs, 96 B of spills
                                // per lane) C3zoom is 5 % and C3 2.6 % faster, but the spills put 66 MB of scratch writes per
                                // 1080p frame on HBM (WRITE_SIZE 91 MB against a 24.9 MB framebuffer): not taken
                                // (profiles/r02/v12_*).  The kernels of transparent scenes have LDS for 12 waves per CU only.
#endif

using namespace rtc;

// =====================================================================================================
//                                              device
// =====================================================================================================
namespace {

// Work counters: [2] ray-triangle tests, [3] ray-primitive tests (one per active lane and leaf visited).
__device__ __forceinline__ void count_tests(const DScene& S, int which, bool active, uint32_t n) {
  if (S.counters) {
    unsigned long long m = __ballot(active ? 1 : 0);
    if (m && (threadIdx.x & 63) == (unsigned)(__ffsll((long long)m) - 1))
      atomicAdd(&S.counters[which], (unsigned long long)__popcll(m) * n);
  }
}

template <bool SHADOW, bool REFR, bool COUNT>
__device__ void sweep(const DScene& S, const Ray& r, bool active, double dist, RadianceState& rs, bool& shadowed,
                      double* s_tri) {
  // World::intersect (world.rs:43-54): every leaf, depth-first.  Group boxes (shape.rs:399-425) gate their
  // descendants exactly as the reference's recursive Group arm does.
  GroupGate gate;
  gate.begin(r);
  for (uint32_t si = 0; si < S.n_segs; ++si) {
    const DSegRec& rec = S.segs[si];
    const DSegment seg = rec.s;
    bool need = gate.reach<false>(S, rec, seg, r, active);
    if (SHADOW) need = need && !shadowed;
    if (seg.is_tri) {
      if (!__syncthreads_or(need ? 1 : 0)) continue;  // no ray of this workgroup reaches the run
      const double* m = rec.xf;
      V3 o{0, 0, 0}, d{0, 0, 0};
      if (need) {  // Ray::transform (ray.rs:19-24) with the cached inverse: once per run, not per triangle
        o = xf_point(m, r.o);
        d = xf_vector(m, r.d);
      }
      for (uint32_t base = 0; base < seg.count; base += TILE_TRIS) {
        const uint32_t n = min((uint32_t)TILE_TRIS, seg.count - base);
        __syncthreads();  // everyone is done with the previous tile
        const double* src = S.tri9 + (size_t)(seg.first_tri + base) * 9;
        for (uint32_t k = threadIdx.x; k < n * 9; k += BLOCK_EXACT) s_tri[k] = src[k];
        __syncthreads();
        if (COUNT) count_tests(S, 2, need, n);
        if (need) {
          for (uint32_t j = 0; j < n; ++j) {
            if (SHADOW) {
              ShadowAcc acc{shadowed, dist};
              intersect_triangle(s_tri + 9 * j, o, d, acc);
              if (shadowed) break;
            } else {
              LeafAcc<REFR> acc(rs, (int)(seg.first_leaf + base + j));
              intersect_triangle(s_tri + 9 * j, o, d, acc);
              acc.end();
            }
          }
        }
      }
    } else if (need) {
      if (COUNT) count_tests(S, 3, true, 1);
      const uint32_t leaf = seg.first_leaf;
      const double* m = rec.xf;
      const V3 o = xf_point(m, r.o);
      const V3 d = xf_vector(m, r.d);
      const uint32_t kind = seg.kind & 0xffu;
      const double* params = S.leaf_params + (size_t)leaf * 3;
      if (SHADOW) {
        ShadowAcc acc{shadowed, dist};
        intersect_primitive(kind, params, o, d, acc);
      } else {
        LeafAcc<REFR> acc(rs, (int)leaf);
        intersect_primitive(kind, params, o, d, acc);
        acc.end();
      }
    }
  }
}

// Region timers of the diagnostics build: one sample per wave (lane 0), shader-clock cycles.
//   6 ray generation, 7 radiance sweep, 8 prepare_computations, 9 shadow sweep, 10 lighting + children, 11 store,
//   12 whole wave
// wave-level loop counters of the diagnostics build: 13 cluster groups, 14 triangle-filter rounds, 15 exact-test rounds,
// 16 triangle-run sweeps (bundles built), 17 of them without a usable bundle axis, 18 tiles visited, 19 per-ray cluster tests
// triangle-run sweeps: 21 ray bundle, 22 exact rounds, 23 filter rounds, 24 cluster-group preparation, 25 the whole run,
// 29 group boxes on the way to the run (GroupGate::reach), 30 object-space ray
#if RTC_REGION_TIMERS == 2
// Mode 2: the timers run in the ordinary (non-counting) kernels too.  A wave adds its samples up in LDS (one-lane ds_add, no
// memory traffic inside the timed regions; mode 1's atomics on one hot line per counter queue the kernel's own loads behind
// them and shift time into whichever region waits for memory first) and adds them to a row of a device array chosen by its
// block when it ends; rtc_debug_counters adds the rows up.  Accelerated kernels only (one wave per workgroup).
//   27 exact rounds: queue entry + record arrival, 28 exact rounds: arithmetic + bookkeeping
constexpr int TIMER_ROWS = 4096;
__device__ unsigned long long g_timers[TIMER_ROWS * 32];
__shared__ unsigned long long s_timers[32];
__device__ __forceinline__ void timers_begin() {
  if ((threadIdx.x & 63) < 32) s_timers[threadIdx.x & 63] = 0ull;
}
__device__ __forceinline__ void timers_end() {
  const uint32_t l = threadIdx.x & 63;
  if (l < 32 && s_timers[l]) atomicAdd(&g_timers[(size_t)(blockIdx.x % TIMER_ROWS) * 32 + l], s_timers[l]);
}
#else
__device__ __forceinline__ void timers_begin() {}
__device__ __forceinline__ void timers_end() {}
#endif
template <bool COUNT>
__device__ __forceinline__ void wave_count(const DScene& S, int which) {
#if RTC_REGION_TIMERS == 2
  if ((threadIdx.x & 63) == 0) atomicAdd(&s_timers[which], 1ull);
#elif RTC_REGION_TIMERS
  if (COUNT && S.counters && (threadIdx.x & 63) == 0) atomicAdd(&S.counters[which], 1ull);
#endif
}
template <bool COUNT>
__device__ __forceinline__ unsigned long long region_clock() {
#if RTC_REGION_TIMERS == 2
  return __builtin_readcyclecounter();
#elif RTC_REGION_TIMERS
  if (COUNT) return __builtin_readcyclecounter();
#endif
  return 0ull;
}
template <bool COUNT>
__device__ __forceinline__ void region_add(const DScene& S, int which, unsigned long long t0, unsigned long long t1) {
#if RTC_REGION_TIMERS == 2
  if ((threadIdx.x & 63) == 0) atomicAdd(&s_timers[which], t1 - t0);
#elif RTC_REGION_TIMERS
  if (COUNT && S.counters && (threadIdx.x & 63) == 0) atomicAdd(&S.counters[which], t1 - t0);
#endif
}

// ---------------------------------------------------------------------------------------------------------
// Per-wave trace state in LDS.  One "slot" = 64 doubles (one per lane).
//   0-2   colour accumulator of the PIXEL this lane owns (other lanes add to it when they trace its rays)
//   3     weight of the lane's current ray
//   4     two u32 per lane: [lane] = remaining | owner << 8 of the current ray, [64 + lane] = accumulation ticket
//   5-10  origin, direction of the current ray (its "home": the registers are free between the sweeps)
//   11    hit parameter during the shadow sweep
//   12-19 (kernels of scenes with a transparent material only) the wave's pool of parked rays: 64 entries of
//         origin, direction, weight, remaining | owner; entries beyond 64 live in global memory (DScene::pool)
constexpr int WS_SLOTS = 12;
#ifndef RTC_POOL_LDS
#define RTC_POOL_LDS 64
#endif
constexpr int WS_POOL_LDS = RTC_POOL_LDS;  // parked rays a wave keeps in LDS (8 doubles each)
// after the slots: 8 doubles for the wave's ray bundle (sweep_accel); kernels of opaque scenes (no pool) also park the
// hit's normal there during the shadow sweep (3 slots) and keep the shadow ray's home behind it (6 slots, see sweep_accel: HOME)
// (both only in kernels with triangle runs: the others have neither a bundle nor a shortage of registers)
constexpr size_t WS_BUNDLE = 8;
constexpr size_t ws_doubles(bool refr, bool tris) {
  return (size_t)WS_SLOTS * 64 + (tris ? WS_BUNDLE : 0) + (refr ? (size_t)WS_POOL_LDS * 8 : (tris ? (size_t)(3 + (RTC_SHADOW_HOME ? 6 : 0)) * 64 : 0));
}
constexpr size_t ws_bundle_at() { return (size_t)WS_SLOTS * 64; }
constexpr size_t ws_pool_at(bool tris) { return (size_t)WS_SLOTS * 64 + (tris ? WS_BUNDLE : 0); }  // pool (REFR) / parked normal

// ---------------------------------------------------------------------------------------------------------
// Accelerated sweep: same intersections as sweep<>() above, fewer tests.  Triangle runs are walked as
// tiles -> clusters -> triangles (rtc_accel.hpp) with wave-uniform loops; each lane applies the conservative
// f32 box / triangle filters and queues the survivors in a private LDS list; the queued (ray, triangle)
// pairs then run the reference's exact f64 Möller–Trumbore (intersect_triangle), all lanes in lock-step, so
// the rare exact tests do not serialise the wave.  Results are bit-identical to the brute-force sweep: the
// filters only drop pairs whose exact test certainly reports nothing, and hit selection / container
// bookkeeping use explicit (t, depth-first leaf index) tie rules, so visiting order is irrelevant.
#ifndef RTC_CAND_CAP
#define RTC_CAND_CAP 8  // 8, 12, 16 and 24 time the same (profiles/r02/v1_variants_cand_cap.txt); 8 leaves LDS for the ray pool
#endif
constexpr int CAND_CAP = RTC_CAND_CAP;  // queued exact tests per lane (LDS: CAND_CAP * BLOCK_ACCEL * 4 B = 2 KiB per wave)
// LDS of an accelerated workgroup, in doubles: the candidate queues (kernels with triangle runs only), then the wave's
// trace state (ws_doubles below): 2 + 7.6 KiB per wave for opaque meshes, 2 + 10.1 KiB with a ray pool
constexpr size_t queue_doubles(bool tris) { return tris ? (size_t)CAND_CAP * BLOCK_ACCEL * 4 / 8 : 0; }

// BACK (radiance rays of transparent scenes only): the sweep that collects the intersections BEHIND the origin on triangle
// runs (LeafAccBack); with RTC_SPLIT_REFR the ordinary radiance sweep of such kernels then only looks forward (LeafAccFwd).
// HOME: the ray also lies in LDS (home[j * 64], j = 0..5: origin, direction of this lane's ray).  During a triangle run the
// registers then hold the OBJECT-space ray (what the exact tests need) and the world ray is read back from its home after the run,
// instead of holding the world ray and rebuilding the object-space one -- twelve loads of the transform and 36 f64 operations --
// at every drain of the exact queue.
template <bool SHADOW, bool REFR, bool COUNT, bool TRIS, bool BACK = false, bool HOME = false>
__device__ void sweep_accel(const DScene& S, const Ray& r_in, bool active, double dist, RadianceState& rs,
                            bool& shadowed, uint32_t* s_cand, double* ws, bool one_origin = false,
                            const __attribute__((address_space(3))) double* home = nullptr) {
  Ray r = r_in;
  lds_vf32* bundle_lds = (lds_vf32*)((__attribute__((address_space(3))) double*)ws + ws_bundle_at());
  GroupGate gate;
  gate.begin(r);
  const uint32_t tid = threadIdx.x;
  uint32_t n_box = 0, n_filter = 0, n_exact = 0, n_prim = 0;
  for (uint32_t si = 0; si < S.n_segs; ++si) {
    const DSegRec& rec = S.segs[si];
    const DSegment seg = TRIS ? ldc(&rec.s) : rec.s;
    const unsigned long long s0 = region_clock<COUNT>();
    bool need = gate.reach<TRIS>(S, rec, seg, r, active);
    if (SHADOW) need = need && !shadowed;
    const unsigned long long s1 = region_clock<COUNT>();
    region_add<COUNT>(S, 29, s0, s1);
    constexpr bool FWD = !SHADOW && !BACK && REFR && TRIS && RTC_SPLIT_REFR;  // forward half of a split radiance sweep
    if (BACK && !seg.is_tri) continue;  // primitives report all their intersections, of either sign, in the forward sweep
    if (TRIS && seg.is_tri) {
      if (__ballot(need ? 1 : 0) == 0) continue;  // no ray of this wave reaches the run
      const double* m = rec.xf;
      RayF rf;
      V3 obj_o{0, 0, 0}, obj_d{0, 0, 1};  // (HOME: live through the run; otherwise dead after the conversion below)
      {
        if (need) {  // Ray::transform (ray.rs:19-24)
          obj_o = xf_point(m, r.o);
          obj_d = xf_vector(m, r.d);
        }
        rf.ox = (float)obj_o.x, rf.oy = (float)obj_o.y, rf.oz = (float)obj_o.z;
        rf.dx = (float)obj_d.x, rf.dy = (float)obj_d.y, rf.dz = (float)obj_d.z;
      }
      {
        const float mo = fmaxf(fmaxf(fabsf(rf.ox), fabsf(rf.oy)), fabsf(rf.oz)) * 1.000001f;
        const float md = fmaxf(fmaxf(fabsf(rf.dx), fabsf(rf.dy)), fabsf(rf.dz)) * 1.000001f;
        rf.D = md;
        rf.PD = (mo + seg.pmax) * md * 1.000001f;
        rf.mb = 1e-5f * (mo + seg.ext);
        if (SHADOW) {
          rf.tlo = 0.f, rf.thi = (float)dist * 1.00001f;
        } else if (BACK) {
          rf.tlo = -INFINITY, rf.thi = 0.f;       // only what lies behind the origin
        } else if (REFR && !FWD) {
          rf.tlo = -INFINITY, rf.thi = INFINITY;  // negative-t intersections feed n1/n2
        } else {
          rf.tlo = 0.f, rf.thi = INFINITY;        // Intersection::hit only looks at t >= 0
        }
      }
      const unsigned long long q0 = region_clock<COUNT>();
      region_add<COUNT>(S, 30, s1, q0);
      unsigned long long q_flush = 0, q_filter = 0, q_group = 0;
#if RTC_REGION_TIMERS == 2
      uint32_t dbg_groups = 0;
#endif
      int ncand = 0;
      auto flush = [&]() {  // exact f64 tests of every queued pair, lanes in lock-step
        const unsigned long long f0 = region_clock<COUNT>();
        // without a home for the world ray the f64 object-space ray is rebuilt here (same expression, same value)
        const V3 o = HOME ? obj_o : xf_point(m, r.o), d = HOME ? obj_d : xf_vector(m, r.d);
        while (__ballot(ncand > 0 ? 1 : 0)) {
          wave_count<COUNT>(S, 15);
#if RTC_REGION_TIMERS == 2
          // diagnostics only: the record is brought into registers and waited for before the arithmetic starts, so that the
          // two halves of a round can be timed separately
          const unsigned long long e0 = region_clock<COUNT>();
          const bool has = ncand > 0;
          double Tl[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
          uint32_t ti = 0;
          int leaf_of = 0;
          if (has) {
            --ncand;
            ti = s_cand[(uint32_t)ncand * BLOCK_ACCEL + tid];
            for (int q = 0; q < 9; ++q) Tl[q] = S.tri9x[(size_t)ti * TRI9X_STRIDE + q];
            if (!SHADOW) leaf_of = (int)(uint32_t)__double_as_longlong(S.tri9x[(size_t)ti * TRI9X_STRIDE + 9]);
          }
          asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
          const unsigned long long e1 = region_clock<COUNT>();
          const double* T = Tl;
          if (has) {
#else
          if (ncand > 0) {
            --ncand;
            const uint32_t ti = s_cand[(uint32_t)ncand * BLOCK_ACCEL + tid];
            // The whole record is requested at once and has arrived before the arithmetic starts.  Left to itself the compiler
            // sinks the loads of p1 and of the leaf index behind the |det| and u tests of intersect_triangle: three dependent
            // round trips to L2 per exact round instead of one.
            const double* Tg = S.tri9x + (size_t)ti * TRI9X_STRIDE;
            double Tl[9];
#pragma unroll
            for (int q = 0; q < 9; ++q) Tl[q] = Tg[q];
            int leaf_of = SHADOW ? 0 : (int)(uint32_t)__double_as_longlong(Tg[9]);
            asm volatile("" : "+v"(Tl[0]), "+v"(Tl[1]), "+v"(Tl[2]), "+v"(Tl[3]), "+v"(Tl[4]), "+v"(Tl[5]), "+v"(Tl[6]), "+v"(Tl[7]),
                         "+v"(Tl[8]), "+v"(leaf_of));
            const double* T = Tl;
#endif
            if (COUNT) ++n_exact;
            if (SHADOW) {
              ShadowAcc acc{shadowed, dist};
              intersect_triangle(T, o, d, acc);
              if (shadowed) ncand = 0;
            } else if (BACK) {
              LeafAccBack acc(rs, leaf_of);
              intersect_triangle(T, o, d, acc);
              acc.end();
            } else if (FWD) {
              LeafAccFwd acc(rs, leaf_of);
              intersect_triangle(T, o, d, acc);
              acc.end();
            } else {
              LeafAcc<REFR> acc(rs, leaf_of);
              intersect_triangle(T, o, d, acc);
              acc.end();
            }
          }
#if RTC_REGION_TIMERS == 2
          region_add<COUNT>(S, 27, e0, e1), region_add<COUNT>(S, 28, e1, region_clock<COUNT>());
#endif
        }
        if (SHADOW) need = need && !shadowed;
        q_flush += region_clock<COUNT>() - f0;
      };
      // tiles and clusters: 64 boxes per step, one per lane, against the wave's ray bundle (bundle_box); only the
      // boxes some ray may hit are then visited with the per-ray test (box_filter) and the triangle filters.
      const uint32_t lane = tid & 63u;
      const uint32_t n_steps = (seg.n_tiles + 63u) / 64u;
#if RTC_TILE_PREFETCH
      // a run of at most 64 tiles (4 096 clusters): its tile boxes are requested before the bundle is built
      DBox tbox0{};
      if (n_steps == 1 && lane < seg.n_tiles) tbox0 = S.tiles[seg.first_tile + lane];
#endif
      int bundle_k;
      bool backwards;
#if RTC_BUNDLE_SGPR
      const Bundle B0 = make_bundle(rf, need, SHADOW);
      bundle_k = B0.k;
      backwards = RTC_FRONT_TO_BACK && ((B0.neg >> ((seg.kind >> 8) & 3u)) & 1u);
#define RTC_BUNDLE() B0
#else
      {
#if RTC_FAST_BUNDLE && RTC_WAVE_REDUCE_DPP
        // one_origin: the rays are the primary rays of a render wave; shadow rays end at the light
        const Bundle B0 = (SHADOW || one_origin) ? make_bundle2(rf, need, SHADOW, true, (seg.kind >> 8) & 3u) : make_bundle(rf, need, SHADOW);
#else
        const Bundle B0 = make_bundle(rf, need, SHADOW);
#endif
        bundle_k = B0.k;
        backwards = RTC_FRONT_TO_BACK && ((B0.neg >> ((seg.kind >> 8) & 3u)) & 1u);
        store_bundle(bundle_lds, B0, lane);
      }
#define RTC_BUNDLE() load_bundle(bundle_lds, bundle_k)
#endif
      const unsigned long long q1 = region_clock<COUNT>();
      wave_count<COUNT>(S, 16);
#if RTC_REGION_TIMERS != 2
      if (bundle_k < 0) wave_count<COUNT>(S, 17);
#endif
      const float tri_pad = RTC_TRI_PAD * seg.ext;
      // How far along the bundle's axis any ray of the wave still looks (wave-uniform).  Radiance sweeps of opaque scenes narrow
      // each ray's range to its nearest hit so far (below); what lies beyond the farthest of those ends cannot be hit by any
      // ray, and is skipped before its boxes are fetched.  Infinite until every ray that takes part has a hit.
      // (kernels of opaque scenes only: the forward sweeps of transparent scenes lose more to the registers it takes than they gain)
      constexpr bool REACH = RTC_REACH_CULL && !SHADOW && !BACK && !REFR && RTC_RADIANCE_EAGER;
      bool kpos = true;
      float reach = INFINITY;
      if (REACH && bundle_k >= 0) {
        const float dk_ = bundle_k == 0 ? rf.dx : bundle_k == 1 ? rf.dy : rf.dz;
        kpos = __ballot((need && dk_ > 0.f) ? 1 : 0) != 0;  // (one sign for all rays on the bundle's axis)
        reach = kpos ? INFINITY : -INFINITY;
      }
      for (uint32_t step = 0; step < n_steps; ++step) {
        const uint32_t tb = (backwards ? n_steps - 1u - step : step) * 64u;
        bool tl = false;
        uint32_t tile_first = 0, tile_count = 0;  // this lane's tile: where its clusters are
        if (tb + lane < seg.n_tiles) {
#if RTC_TILE_PREFETCH
          const DBox tbox = n_steps == 1 ? tbox0 : S.tiles[seg.first_tile + tb + lane];
#else
          const DBox tbox = S.tiles[seg.first_tile + tb + lane];
#endif
          tile_first = tbox.first, tile_count = tbox.count;
          tl = bundle_box(tbox, RTC_BUNDLE());
        }
        if (COUNT) ++n_box;
        unsigned long long tmask = __ballot(tl ? 1 : 0);
        while (tmask) {
          // roughly front to back: records ascend along the run's first split axis, so a wave whose rays all travel
          // in the negative direction on that axis takes tiles and clusters from the far end of the masks
          const uint32_t tsel = backwards ? 63u - (uint32_t)__clzll((long long)tmask) : (uint32_t)__ffsll((long long)tmask) - 1u;
          tmask &= ~(1ull << tsel);
          if (REACH && bundle_k >= 0 && fabsf(reach) < INFINITY) {  // does the whole tile lie beyond every ray's nearest hit?
            // (its box is read again, by a scalar load, only once there is a finite reach: no register held for it meanwhile)
            const DBox* tb_ = S.tiles + (seg.first_tile + tb + tsel);
            const float tn = kpos ? (bundle_k == 0 ? tb_->mn[0] : bundle_k == 1 ? tb_->mn[1] : tb_->mn[2])
                                  : (bundle_k == 0 ? tb_->mx[0] : bundle_k == 1 ? tb_->mx[1] : tb_->mx[2]);
            if (kpos ? tn > reach : tn < reach) continue;
          }
          // the selected tile's header comes from the lane that tested its box (no second, dependent load)
          struct { uint32_t first, count; } T;
          T.first = (uint32_t)__builtin_amdgcn_readlane((int)tile_first, (int)tsel);
          T.count = (uint32_t)__builtin_amdgcn_readlane((int)tile_count, (int)tsel);
          wave_count<COUNT>(S, 18);
          bool cl = false;
          DBox cbox{};
          if (lane < T.count) {
            cbox = S.clusters[T.first + lane];
            cl = bundle_box(cbox, RTC_BUNDLE());
          }
          if (COUNT) ++n_box;
          unsigned long long cmask = __ballot(cl ? 1 : 0);
          if (RTC_BLOCK_FILTER && __popcll(cmask) >= RTC_BLOCK_FILTER) {
            // Many clusters survived the bundle test (an incoherent wave): test the tile's eight blocks of eight
            // clusters per ray first and drop the blocks no ray can hit, instead of testing their clusters one by one.
            const DBox* blk = S.blocks + (size_t)(seg.first_tile + tb + tsel) * 8;
#pragma unroll
            for (uint32_t k = 0; k < 8; ++k) {
              if ((cmask >> (8 * k)) & 0xffull) {
                if (COUNT && need) ++n_box;
                const bool bh = (RTC_BRANCH_FREE_ROUNDS & 4) ? (box_filter(RTC_BLK_LDC ? ldc(&blk[k]) : blk[k], rf) & need)
                                                       : (need && box_filter(RTC_BLK_LDC ? ldc(&blk[k]) : blk[k], rf));
                if (__ballot(bh ? 1 : 0) == 0) cmask &= ~(0xffull << (8 * k));
              }
            }
            cl = cl && ((cmask >> lane) & 1ull);
          }
          // Triangle-level wave culling.  The surviving clusters are taken SLOTS at a time; every lane rebuilds the
          // box of ONE of their triangles from its filter record and tests it against the bundle.  The triangles some
          // ray may hit are then broadcast from the lane that holds them (v_readlane -> SGPR operands, no memory
          // access) and filtered per ray.  For the coherent waves of primary and shadow rays this leaves a handful of
          // filter evaluations per cluster instead of CLUSTER_TRIS.
          constexpr uint32_t SLOTS = 64 / CLUSTER_TRIS;
          const uint32_t n_cl = (uint32_t)__popcll(cmask);
          // compaction: the lane of the k-th surviving cluster sends its number to lane k (the others fill the tail)
          const uint32_t below =
              __builtin_amdgcn_mbcnt_hi((uint32_t)(cmask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)cmask, 0u));
          const uint32_t rank = backwards ? n_cl - 1u - below : below;  // position among the surviving clusters
          const uint32_t dest = cl ? rank : n_cl + (lane - below);
          const uint32_t compact = (uint32_t)__builtin_amdgcn_ds_permute((int)(dest << 2), (int)lane);
          for (uint32_t g0 = 0; g0 < n_cl; g0 += SLOTS) {
            wave_count<COUNT>(S, 13);
#if RTC_REGION_TIMERS == 2
            ++dbg_groups;
#endif
            const unsigned long long g0c = region_clock<COUNT>();
            // this lane's triangle of the group: the record is requested first, so that its latency overlaps the
            // per-ray cluster tests below
            const uint32_t slot = g0 + lane / CLUSTER_TRIS;
            const uint32_t csel = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((slot & 63u) << 2), (int)compact);
            const uint32_t rel = (T.first - seg.first_cluster + csel) * CLUSTER_TRIS + lane % CLUSTER_TRIS;
            const uint32_t rec = seg.first_rec + rel;
            const bool mine_ok = slot < n_cl && rel < seg.count;
            DTriF mine{};
            if (mine_ok) mine = S.trif[rec];
            // per-ray test of the group's cluster boxes (broadcast from the lanes that hold them): a cluster no
            // ray can hit is dropped whole, which is what keeps incoherent waves (wide bundles) cheap
            uint32_t alive = 0;
            for (uint32_t i = 0; i < SLOTS && g0 + i < n_cl; ++i) {
              const int bl = __builtin_amdgcn_readlane((int)compact, (int)(g0 + i));
              DBox Cb;
#pragma unroll
              for (int a = 0; a < 3; ++a) {
                Cb.mn[a] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cbox.mn[a]), bl));
                Cb.mx[a] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cbox.mx[a]), bl));
              }
              if (COUNT && need) ++n_box;
              wave_count<COUNT>(S, 19);
              const bool ch = (RTC_BRANCH_FREE_ROUNDS & 2) ? (box_filter(Cb, rf) & need) : (need && box_filter(Cb, rf));
              if (__ballot(ch ? 1 : 0)) alive |= 1u << i;
            }
            if (!alive) {
              q_group += region_clock<COUNT>() - g0c;
              continue;
            }
            bool th = false;
            if (mine_ok && ((alive >> (lane / CLUSTER_TRIS)) & 1u)) th = bundle_tri(mine, tri_pad, RTC_BUNDLE());
            if (COUNT) ++n_box;
            unsigned long long trmask = __ballot(th ? 1 : 0);
            const unsigned long long g1c = region_clock<COUNT>();
            q_group += g1c - g0c;
            auto bcast = [&](int src) {  // one lane's triangle record -> SGPR operands of every lane
              DTriF t;
#pragma unroll
              for (int a = 0; a < 3; ++a) {
                t.p1[a] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine.p1[a]), src));
                t.e1[a] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine.e1[a]), src));
                t.e2[a] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine.e2[a]), src));
              }
              t.c1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine.c1), src));
              t.c2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine.c2), src));
              t.c3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine.c3), src));
              return t;
            };
            while (trmask) {
              const int src = __ffsll((long long)trmask) - 1;
              trmask &= trmask - 1ull;
              const DTriF t = bcast(src);
              const uint32_t trec = (uint32_t)__builtin_amdgcn_readlane((int)rec, src);
              if (__ballot(ncand >= CAND_CAP ? 1 : 0)) flush();
              wave_count<COUNT>(S, 14);
              if (COUNT && need) ++n_filter;
#if RTC_BRANCH_FREE_ROUNDS & 1
              // no exec-mask branch per round: every lane evaluates the filter and writes the slot its next queue entry would take
              // (ncand < CAND_CAP here); only a pass makes the entry count
              s_cand[(uint32_t)ncand * BLOCK_ACCEL + tid] = trec;
              ncand += (tri_filter(t, rf) & need) ? 1 : 0;
#else
              if (need && tri_filter(t, rf)) s_cand[(uint32_t)(ncand++) * BLOCK_ACCEL + tid] = trec;
#endif
            }
            q_filter += region_clock<COUNT>() - g1c;
            if (SHADOW && RTC_SHADOW_EAGER == 2 && __ballot(ncand > 0 ? 1 : 0)) {
              flush();
              if (__ballot(need ? 1 : 0) == 0) goto run_done;
            }
            if (!SHADOW && !BACK && ((REFR && !FWD) ? RTC_REFR_EAGER : RTC_RADIANCE_EAGER) && __ballot(ncand > 0 ? 1 : 0)) {
              flush();
              // Intersection::hit keeps the lowest non-negative t (ties: lower leaf index): nothing beyond the best
              // hit so far can win.  Widened upwards, so a tie at exactly best_t is still examined.
              rf.thi = fminf(rf.thi, (float)rs.best_t * 1.00001f);
              if (REACH && bundle_k >= 0) {
                // a ray's far end along the axis, moved outwards by twice its box margin (>= the bundle's own margin: the anchor
                // is the origin here, so 1e-5 * max|o| <= rf.mb); rays that do not take part leave the reach alone
                const float ok_ = bundle_k == 0 ? rf.ox : bundle_k == 1 ? rf.oy : rf.oz;
                const float dk_ = bundle_k == 0 ? rf.dx : bundle_k == 1 ? rf.dy : rf.dz;
                const float far_ = fmaf(dk_, rf.thi, ok_);
                if (kpos) reach = wave_max(need ? far_ + 2.f * rf.mb : -INFINITY);
                else reach = wave_min(need ? far_ - 2.f * rf.mb : INFINITY);
                if (lane == 0) bundle_lds[kpos ? 11 : 10] = reach;  // the bundle's extent along the axis: cluster and triangle boxes
              }
              // With a transparent material the intersections behind the origin matter too, but only the two leaves
              // with the greatest negative t (RadianceState: the end of prepare_computations' container list): a
              // triangle behind the second of them cannot displace either.  Widened downwards (ties at neg_t2 go to
              // the later leaf and are still examined); -inf until two such leaves are known.
              if (REFR && !FWD) rf.tlo = fmaxf(rf.tlo, (float)rs.neg_t2 * 1.00001f);
            }
          }
          if (SHADOW && RTC_SHADOW_EAGER == 1 && __ballot(ncand > 0 ? 1 : 0)) {
            flush();
            if (__ballot(need ? 1 : 0) == 0) goto run_done;
          }
        }
      }
      if (__ballot(ncand > 0 ? 1 : 0)) flush();  // (the drain rebuilds the f64 object-space ray first: not for an empty queue)
    run_done:;
      if (HOME) {  // the world ray again (the gates and transforms of the segments that follow)
        r.o = {home[0], home[64], home[128]};
        r.d = {home[192], home[256], home[320]};
      }
#undef RTC_BUNDLE
      {
        const unsigned long long q2 = region_clock<COUNT>();
        region_add<COUNT>(S, 21, q0, q1), region_add<COUNT>(S, 25, q0, q2);
#if RTC_REGION_TIMERS == 2
        if (dbg_groups == 0) wave_count<COUNT>(S, 17), region_add<COUNT>(S, 31, q0, q2);  // runs that ended above the cluster groups
#endif
        region_add<COUNT>(S, 22, 0ull, q_flush), region_add<COUNT>(S, 23, 0ull, q_filter), region_add<COUNT>(S, 24, 0ull, q_group);
      }
    } else {
      if (__ballot(need ? 1 : 0) == 0) continue;  // no ray of this wave reaches the leaf
      const uint32_t leaf = seg.first_leaf;
      const double* m = rec.xf;
      V3 o{0, 0, 0}, d{0, 0, 1};
      if (need) {  // Ray::transform (ray.rs:19-24)
        o = xf_point(m, r.o);
        d = xf_vector(m, r.d);
      }
      const uint32_t kind = seg.kind & 0xffu;
      bool go = need;
      if (RTC_CUBE_FILTER && kind == K_CUBE) {  // f32 bounds test in front of the six f64 divisions of the Cube arm
        const float tlo = (SHADOW || !REFR) ? 0.f : -INFINITY;  // negative t only feeds n1/n2 (see the triangle runs)
        const float thi = SHADOW ? (float)dist * 1.00001f : INFINITY;
        if (COUNT && need) ++n_box;
        go = need && cube_maybe(o, d, tlo, thi);
        if (__ballot(go ? 1 : 0) == 0) continue;
      }
      if (go) {
        if (COUNT) ++n_prim;
        const double* params = S.leaf_params + (size_t)leaf * 3;
        if (SHADOW) {
          ShadowAcc acc{shadowed, dist};
          intersect_primitive(kind, params, o, d, acc);
        } else {
          LeafAcc<REFR> acc(rs, (int)leaf);
          intersect_primitive(kind, params, o, d, acc);
          acc.end();
        }
      }
    }
  }
  if (COUNT && S.counters) {
    if (n_exact) atomicAdd(&S.counters[2], (unsigned long long)n_exact);
    if (n_prim) atomicAdd(&S.counters[3], (unsigned long long)n_prim);
    if (n_filter) atomicAdd(&S.counters[4], (unsigned long long)n_filter);
    if (n_box) atomicAdd(&S.counters[5], (unsigned long long)n_box);
  }
}

__device__ __forceinline__ void count_rays(const DScene& S, int which, bool active) {
  if (S.counters) {
    unsigned long long m = __ballot(active ? 1 : 0);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(&S.counters[which], (unsigned long long)__popcll(m));
  }
}

// Computations::schlick (intersection.rs:107-128)
__device__ __forceinline__ double schlick(double cosv, double n1, double n2) {
  if (n1 > n2) {
    double n = n1 / n2;
    double sin2_t = (n * n) * (1.0 - cosv * cosv);
    if (sin2_t > 1.0) return 1.0;
    cosv = sqrt(1.0 - sin2_t);
  }
  double q = (n1 - n2) / (n1 + n2);
  double r0 = q * q;
  double x = 1.0 - cosv;
  double x2 = x * x;
  double x4 = x2 * x2;
  return r0 + (1.0 - r0) * (x * x4);  // powi(5)
}

// World::color_at (world.rs:80-163) for one ray per thread, cooperatively across the workgroup.
// `limit` is RECURSION_LIMIT.  Every thread of the block must call this (barriers inside the brute-force sweep).
// ACCEL selects sweep_accel (no workgroup barriers) over the brute-force LDS sweep; COUNT adds the work counters
// (never set in timed launches).  `ws` = this wave's state region (above), `wave_id` = its number in the launch.
//
// The reference's recursion (reflected_color / refracted_color inside shade_hit) becomes a loop over rays.  A hit
// spawns at most two children.  One continues in the lane that traced the parent.  The other one (only a material that
// is reflective AND transparent makes two) goes to a lane of the wave that has run out of work, or, when every lane is
// busy, to the wave's pool of parked rays, from which lanes that run out of work later refill (wavefront ballot /
// prefix sums: live rays are kept compacted on the lanes, the north-star's item).  A lane may therefore trace rays of
// another lane's pixel: every ray carries its owner, and its contribution weight * surface is added to the owner's
// accumulator.  The order of those additions must not depend on the sweep (the accelerated and the brute-force kernels
// have to agree bit for bit) nor on hardware arbitration: which lane holds which ray is a function of the rays' hit /
// miss history only, identical for both sweeps, and when several lanes hold contributions for one pixel in the same
// round they add them in ascending lane order (an integer ds_min ticket per owner decides whose turn it is).
template <bool REFR, bool ACCEL, bool COUNT, bool TRIS, bool POOLG>
__device__ void trace(const DScene& S, const Ray& primary, bool valid, int limit, double rgb[3], double* s_tri, double* ws,
                      uint32_t wave_id, bool one_origin = false) {
  const uint32_t ln = threadIdx.x & 63u;
  // explicit LDS pointers (32-bit, ds_* instructions): volatile generic pointers would become flat accesses
  // Only what another lane reads or writes is volatile: the accumulators, the tickets and the pool.  A lane's home
  // (slots 3-11) is private to it, so the compiler may merge and schedule those accesses freely.
  typedef __attribute__((address_space(3))) volatile double lds_f64;
  typedef __attribute__((address_space(3))) volatile uint32_t lds_u32;
  typedef __attribute__((address_space(3))) double lds_pf64;
  typedef __attribute__((address_space(3))) uint32_t lds_pu32;
  lds_f64* wsv = (lds_f64*)ws;
  lds_pf64* my = (lds_pf64*)ws + ln;  // my[k * 64] = slot k of this lane
  lds_pu32* meta = (lds_pu32*)((lds_pf64*)ws + 4 * 64);
  lds_u32* ticket = (lds_u32*)meta + 64;
  uint32_t* s_cand = reinterpret_cast<uint32_t*>(s_tri);
  wsv[ln] = 0.0, wsv[64 + ln] = 0.0, wsv[128 + ln] = 0.0;
  bool have = valid && limit >= 1;  // internal_color_at: remaining < 1 -> BLACK (world.rs:85-87)
  if (have) {
    my[5 * 64] = primary.o.x, my[6 * 64] = primary.o.y, my[7 * 64] = primary.o.z;
    my[8 * 64] = primary.d.x, my[9 * 64] = primary.d.y, my[10 * 64] = primary.d.z;
    my[3 * 64] = 1.0;
    meta[ln] = (uint32_t)limit | (ln << 8);
  }
  uint32_t pool_n = 0;  // rays in the wave's pool (wave-uniform)
  bool mixed = false;   // some lane may hold a ray of another lane's pixel (wave-uniform: set by the first refill)
  bool first_round = one_origin;  // the rays of this round are a camera's primary rays: they share their origin (wave-uniform)
  while (ACCEL ? (__ballot(have ? 1 : 0) != 0) : (__syncthreads_or(have ? 1 : 0) != 0)) {
    const bool active = have;
    Ray r{{0, 0, 0}, {0, 0, 1}};
    if (active) {
      r.o = {my[5 * 64], my[6 * 64], my[7 * 64]};
      r.d = {my[8 * 64], my[9 * 64], my[10 * 64]};
    }
    have = false;
    if (COUNT) count_rays(S, 0, active);
    RadianceState rs;
    // NO_LEAF is greater than every leaf index, so an intersection at t = +inf (a cube or group-free slab hit by a
    // ray whose direction components are all below EPSILON: check_axis yields -inf/+inf, shape.rs:593-599) still
    // wins the (t, leaf) comparison against the initial state, as Intersection::hit accepts it (t >= 0).
    rs.best_t = INFINITY, rs.best_leaf = NO_LEAF, rs.best_parity = 0;
    rs.neg_t1 = rs.neg_t2 = -INFINITY, rs.neg_leaf1 = rs.neg_leaf2 = -1;
    rs.cflags = S.leaf_cflags;
    bool shadowed = false;
    const unsigned long long c0 = region_clock<COUNT>();
    if (ACCEL)
      sweep_accel<false, REFR, COUNT, TRIS, false, TRIS>(S, r, active, 0.0, rs, shadowed, s_cand, ws, first_round, my + 5 * 64);
    else
      sweep<false, REFR, COUNT>(S, r, active, 0.0, rs, shadowed, s_tri);
    first_round = false;
    const bool has_hit = active && rs.best_leaf != NO_LEAF;
    const unsigned long long c1 = region_clock<COUNT>();

    // Intersection::prepare_computations (intersection.rs:17-77).  Only the normal survives the shadow sweep in
    // registers; point / eyev / over_point / material are rebuilt after it from the ray (read back from its home in
    // LDS) and the hit (same expressions, same values) so that they are not spilled around the sweep.
    V3 normalv{0, 1, 0};
    Ray sray{{0, 0, 0}, {0, 0, 1}};
    double distance = 0.0;
    bool want_shadow = has_hit;  // does shade_hit need the answer of is_shadowed for this hit?
    if (has_hit) {
      const int leaf = rs.best_leaf;
      const uint32_t kind = S.leaf_kind[leaf];
      const V3 point = r.o + r.d * rs.best_t;  // Ray::position
      const V3 eyev = -r.d;
      if (TRIS && kind == K_TRIANGLE) {
        const double* nw = S.leaf_nw + (size_t)leaf * 3;  // by leaf: independent of the kind load above
        normalv = {nw[0], nw[1], nw[2]};
      } else {
        normalv = normal_at_primitive(kind, S.leaf_params + (size_t)leaf * 3, S.xf + (size_t)S.leaf_xf[leaf] * 12, point);
      }
      if (dot(normalv, eyev) < 0.0) normalv = -normalv;  // inside
      const V3 over_point = point + normalv * EPSILON;
      // World::is_shadowed (world.rs:100-114)
      V3 v = V3{S.light_pos[0], S.light_pos[1], S.light_pos[2]} - over_point;
      distance = magnitude(v);
      sray.o = over_point;
      sray.d = normalize(v);
      my[11 * 64] = rs.best_t;
#if RTC_SKIP_UNUSED_SHADOW
      // Material::lighting consults in_shadow only when light_dot_normal >= 0 (material.rs:57-60): a hit that faces away
      // from the light is black in diffuse and specular whatever is_shadowed says.  The reference still casts the ray (it is
      // counted below); its sweep is skipped.  Same expressions as lighting(): lightv from `point`, not over_point.
      // The sign of dot(normalize(l), n) is the sign of dot(l, n) unless that is within rounding noise of zero
      // (|difference| < 1e-15 * sum |l_i n_i| / |l|); only then is lighting()'s own expression evaluated here.
      const V3 lv = V3{S.light_pos[0], S.light_pos[1], S.light_pos[2]} - point;
      const double d0 = dot(lv, normalv);
      const double band = 1e-12 * (fabs(lv.x * normalv.x) + fabs(lv.y * normalv.y) + fabs(lv.z * normalv.z));
      if (d0 > band)
        want_shadow = true;
      else if (d0 < -band)
        want_shadow = false;
      else
        want_shadow = dot(normalize(lv), normalv) >= 0.0;  // includes NaN / zero-length cases: exactly what lighting() tests
#endif
    }
    if (COUNT) count_rays(S, 1, has_hit), count_rays(S, 26, has_hit && !want_shadow);
    const unsigned long long c2 = region_clock<COUNT>();
    if (ACCEL) {  // waves do not meet at barriers here: one without a hit skips the shadow sweep altogether
      if (__ballot(want_shadow ? 1 : 0)) {
        // opaque scenes: the normal waits in LDS during the shadow sweep (its six registers are what the sweep is short of)
        lds_pf64* npark = (lds_pf64*)ws + ws_pool_at(true) + ln;
        constexpr bool SHOME = RTC_SHADOW_HOME && !REFR && TRIS;
        if (!REFR && TRIS) {
          npark[0] = normalv.x, npark[64] = normalv.y, npark[128] = normalv.z;
          normalv = V3{0, 1, 0};
          if (SHOME) {
            lds_pf64* sh = npark + 3 * 64;  // the shadow ray's home
            sh[0] = sray.o.x, sh[64] = sray.o.y, sh[128] = sray.o.z, sh[192] = sray.d.x, sh[256] = sray.d.y, sh[320] = sray.d.z;
          }
        }
        sweep_accel<true, false, COUNT, TRIS, false, SHOME>(S, sray, want_shadow, distance, rs, shadowed, s_cand, ws, false, npark + 3 * 64);
        if (!REFR && TRIS) normalv = V3{npark[0], npark[64], npark[128]};
      }
    } else
      sweep<true, false, COUNT>(S, sray, want_shadow, distance, rs, shadowed, s_tri);
    const unsigned long long c3 = region_clock<COUNT>();

    // n1 / n2 (intersection.rs:27-62) need what the ray's line meets BEHIND its origin; the forward radiance sweep of these
    // kernels did not look there on triangle runs.  Only a transparent hit that still has a generation to spawn uses them.
    if (ACCEL && REFR && TRIS && RTC_SPLIT_REFR) {
      bool need_back = false;
      if (has_hit) need_back = S.mats[S.leaf_mat[rs.best_leaf]].transparency != 0.0 && (int)(meta[ln] & 0xffu) - 3 >= 1;
      if (__ballot(need_back ? 1 : 0)) {
        Ray rb{{0, 0, 0}, {0, 0, 1}};
        if (need_back) {
          rb.o = {my[5 * 64], my[6 * 64], my[7 * 64]};
          rb.d = {my[8 * 64], my[9 * 64], my[10 * 64]};
        }
        bool unused = false;
        sweep_accel<false, true, COUNT, TRIS, true>(S, rb, need_back, 0.0, rs, unused, s_cand, ws);
      }
    }
    // World::shade_hit (world.rs:56-78) and the children: reflected_color (world.rs:116-129), refracted_color (:131-163)
    bool has_c1 = false, has_c2 = false;  // c1 continues in this lane; c2 is the second child of a hit that has two
    V3 c1_o{0, 0, 0}, c1_d{0, 0, 0}, c2_o{0, 0, 0}, c2_d{0, 0, 0};
    double c1_k = 0.0, c2_k = 0.0, surface[3] = {0.0, 0.0, 0.0};
    uint32_t child_meta = 0;
    const uint32_t cur_meta = meta[ln];
    const uint32_t own = REFR ? (cur_meta >> 8) : ln;
    if (has_hit) {
      const int leaf = rs.best_leaf;
      const double* inv = S.xf + (size_t)S.leaf_xf[leaf] * 12;
      const DMaterial* mat = &S.mats[S.leaf_mat[leaf]];
      const double hit_t = my[11 * 64];
      Ray rr;  // the ray that was traced, read back from its home (same values)
      rr.o = {my[5 * 64], my[6 * 64], my[7 * 64]};
      rr.d = {my[8 * 64], my[9 * 64], my[10 * 64]};
      const V3 point = rr.o + rr.d * hit_t;
      const V3 eyev = -rr.d;
      const V3 over_point = point + normalv * EPSILON;
      lighting(S, *mat, inv, point, eyev, normalv, shadowed, surface);
      const double w = my[3 * 64];
      surface[0] = surface[0] * w, surface[1] = surface[1] * w, surface[2] = surface[2] * w;
      // remaining: internal_color_at(rem) -> shade_hit(rem-1) -> reflected/refracted_color(rem-2)
      //            -> internal_color_at(rem-3).  Children exist only if rem-3 >= 1.
      const int child_rem = (int)(cur_meta & 0xffu) - 3;
      const bool reflective = mat->reflective != 0.0;    // world.rs:121 (== 0.0 -> BLACK)
      const bool transparent = mat->transparency != 0.0;  // world.rs:136
      if (child_rem >= 1 && (reflective || transparent)) {
        child_meta = (uint32_t)child_rem | (own << 8);
        const bool both = mat->reflective > 0.0 && mat->transparency > 0.0;  // world.rs:71
        double n1 = 1.0, n2 = 1.0;
        if (REFR && transparent) {
          // containers list just before / after the hit (intersection.rs:27-62), see RadianceState
          if (rs.neg_leaf1 >= 0) n1 = S.mats[S.leaf_mat[rs.neg_leaf1]].refractive_index;
          if (rs.best_parity) {  // the hit object was in the list: it is removed
            const bool same = S.leaf_class ? (rs.neg_leaf1 >= 0 && S.leaf_class[rs.neg_leaf1] == S.leaf_class[rs.best_leaf])
                                           : (rs.neg_leaf1 == rs.best_leaf);
            const int other = same ? rs.neg_leaf2 : rs.neg_leaf1;
            if (other >= 0) n2 = S.mats[S.leaf_mat[other]].refractive_index;
          } else {  // it is pushed and becomes the last element
            n2 = mat->refractive_index;
          }
        }
        const double cos_i = dot(eyev, normalv);
        double reflectance = 1.0;
        if (both) reflectance = schlick(cos_i, n1, n2);
        bool has_refl = false, has_refr = false;
        V3 refl_o{0, 0, 0}, refl_d{0, 0, 0}, refr_o{0, 0, 0}, refr_d{0, 0, 0};
        double refl_k = 0.0, refr_k = 0.0;
        if (reflective) {
          refl_o = over_point, refl_d = reflect(rr.d, normalv);
          refl_k = w * mat->reflective;
          if (both) refl_k = refl_k * reflectance;
          has_refl = true;
        }
        if (transparent) {
          double n_ratio = n1 / n2;
          double sin2_t = (n_ratio * n_ratio) * (1.0 - cos_i * cos_i);
          if (!(sin2_t > 1.0)) {  // else total internal reflection: BLACK
            double cos_t = sqrt(1.0 - sin2_t);
            refr_d = normalv * (n_ratio * cos_i - cos_t) - eyev * n_ratio;
            refr_o = point - normalv * EPSILON;  // under_point
            refr_k = w * mat->transparency;
            if (both) refr_k = refr_k * (1.0 - reflectance);
            has_refr = true;
          }
        }
#if RTC_PARK_REFRACTED
        // When there are two, the REFLECTED child continues here and the refracted one is parked: depth-first into the child that
        // does not branch again (it leaves the glass; the refracted one meets glass next and has two children itself), so that
        // the parked rays of a wave peak at about 64 instead of 64 per generation.
        if (has_refl) {
          has_c1 = true, c1_o = refl_o, c1_d = refl_d, c1_k = refl_k;
          if (REFR && has_refr) has_c2 = true, c2_o = refr_o, c2_d = refr_d, c2_k = refr_k;
        } else if (has_refr) {
          has_c1 = true, c1_o = refr_o, c1_d = refr_d, c1_k = refr_k;
        }
#else
        // the refracted child continues here when there are two
        if (has_refr) {
          has_c1 = true, c1_o = refr_o, c1_d = refr_d, c1_k = refr_k;
          if (REFR && has_refl) has_c2 = true, c2_o = refl_o, c2_d = refl_d, c2_k = refl_k;
        } else if (has_refl) {
          has_c1 = true, c1_o = refl_o, c1_d = refl_d, c1_k = refl_k;
        }
#endif
      }
    }
    // ---- the pixel's colour: sum of weight * surface over its rays --------------------------------------------
    if (!REFR) {  // a lane only ever traces rays of its own pixel
      if (has_hit) wsv[ln] += surface[0], wsv[64 + ln] += surface[1], wsv[128 + ln] += surface[2];
    } else {
      // Do two lanes hold contributions for the same pixel?  Every lane posts its number in its owner's ticket word
      // (whichever store wins) and reads it back: some lane reads another's number iff an owner occurs twice.  The
      // common case (none) adds directly; otherwise the additions go by rounds, lowest lane first, decided by an integer
      // ds_min per owner, so the order of the f64 additions never depends on how the hardware arbitrates a store.
      bool pend = has_hit;
      bool lost = false;
      if (mixed) {  // wave-uniform: until the first refill every lane traces rays of its own pixel only
        if (pend) ticket[own] = ln;
        lost = pend && ticket[own] != ln;
      }
      if (__ballot(lost ? 1 : 0) == 0) {
        if (pend) wsv[own] += surface[0], wsv[64 + own] += surface[1], wsv[128 + own] += surface[2];
      } else {
        while (__ballot(pend ? 1 : 0)) {
          if (pend) ticket[own] = 64u;
          if (pend) __hip_atomic_fetch_min((lds_pu32*)&ticket[own], ln, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if (pend && ticket[own] == ln) {
            wsv[own] += surface[0], wsv[64 + own] += surface[1], wsv[128 + own] += surface[2];
            pend = false;
          }
        }
      }
    }
    // ---- next round's rays ---------------------------------------------------------------------------------
    if (has_c1) {
      my[5 * 64] = c1_o.x, my[6 * 64] = c1_o.y, my[7 * 64] = c1_o.z;
      my[8 * 64] = c1_d.x, my[9 * 64] = c1_d.y, my[10 * 64] = c1_d.z;
      my[3 * 64] = c1_k;
      meta[ln] = child_meta;
      have = true;
    }
    if (REFR) {
      // Second children are parked in the wave's pool (compact positions from a ballot / prefix sum); lanes without
      // work refill from it, newest first, once fewer than RTC_REFILL_BELOW lanes are still busy.  Refilling in batches
      // keeps rays of one kind (the reflected children of neighbouring pixels) together in a wave, which is what the
      // wave-level culling of the sweeps lives on; refilling lane by lane (threshold 64) mixes them and costs more box
      // tests than the fuller lanes save (profiles/r02/v2_variants_refill_threshold.txt).
      const unsigned long long extra_m = __ballot(has_c2 ? 1 : 0);
      lds_f64* pool = wsv + ws_pool_at(ACCEL ? TRIS : true);
      if (extra_m) {  // wave-uniform
        const uint32_t extra_rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(extra_m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)extra_m, 0u));
        if (has_c2) {
          const uint32_t e = pool_n + extra_rank;
          if (!POOLG || e < (uint32_t)WS_POOL_LDS) {
            pool[0 * WS_POOL_LDS + e] = c2_o.x, pool[1 * WS_POOL_LDS + e] = c2_o.y, pool[2 * WS_POOL_LDS + e] = c2_o.z;
            pool[3 * WS_POOL_LDS + e] = c2_d.x, pool[4 * WS_POOL_LDS + e] = c2_d.y, pool[5 * WS_POOL_LDS + e] = c2_d.z;
            pool[6 * WS_POOL_LDS + e] = c2_k;
            ((lds_u32*)(pool + 7 * WS_POOL_LDS))[e] = child_meta;
          } else if (POOLG && e - WS_POOL_LDS < S.pool_cap) {
            // (the wave's base address is rebuilt here, from a value the optimiser cannot see through: hoisted out of the
            // trace loop it would be spilled to scratch by every wave for a path few of them take)
            uint32_t wid = wave_id;
            asm volatile("" : "+v"(wid));
            double* g = S.pool + (size_t)wid * S.pool_cap * 8 + (e - WS_POOL_LDS);
            g[0] = c2_o.x, g[S.pool_cap] = c2_o.y, g[2 * (size_t)S.pool_cap] = c2_o.z;
            g[3 * (size_t)S.pool_cap] = c2_d.x, g[4 * (size_t)S.pool_cap] = c2_d.y, g[5 * (size_t)S.pool_cap] = c2_d.z;
            g[6 * (size_t)S.pool_cap] = c2_k;
            g[7 * (size_t)S.pool_cap] = __longlong_as_double((long long)child_meta);
          } else if (POOLG) {
            atomicOr(S.pool_overflow, 1u);  // more rays parked than the launch provided for: the ray is dropped, the host sees
          }                                 // the flag, enlarges the pools and renders the frame again (check_pool_flag)
        }
        pool_n += (uint32_t)__popcll(extra_m);
      }
      if (pool_n) {  // wave-uniform
        const unsigned long long idle_m = __ballot(have ? 0 : 1);
        const uint32_t n_idle = (uint32_t)__popcll(idle_m);
        if (64u - n_idle < (uint32_t)RTC_REFILL_BELOW) {
          // (taken from a copy the optimiser cannot see through: with a threshold of 1 the mask is all ones here, the rank is the
          // lane number, and hoisted out of the trace loop it would be spilled to scratch by every wave)
          unsigned long long idle_m2 = idle_m;
          if (ACCEL && TRIS) asm volatile("" : "+s"(idle_m2));  // (the mesh kernels are the ones short of registers; C2's lose 7 % with it)
          const uint32_t idle_rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle_m2 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle_m2, 0u));
          const uint32_t k = min(n_idle, pool_n);
          if (!have && idle_rank < k) {
            const uint32_t e = pool_n - 1u - idle_rank;
            if (!POOLG || e < (uint32_t)WS_POOL_LDS) {
              my[5 * 64] = pool[0 * WS_POOL_LDS + e], my[6 * 64] = pool[1 * WS_POOL_LDS + e], my[7 * 64] = pool[2 * WS_POOL_LDS + e];
              my[8 * 64] = pool[3 * WS_POOL_LDS + e], my[9 * 64] = pool[4 * WS_POOL_LDS + e], my[10 * 64] = pool[5 * WS_POOL_LDS + e];
              my[3 * 64] = pool[6 * WS_POOL_LDS + e];
              meta[ln] = ((lds_u32*)(pool + 7 * WS_POOL_LDS))[e];
              have = true;
            } else if (POOLG && e - WS_POOL_LDS < S.pool_cap) {
              uint32_t wid = wave_id;
              asm volatile("" : "+v"(wid));
              const double* g = S.pool + (size_t)wid * S.pool_cap * 8 + (e - WS_POOL_LDS);
              my[5 * 64] = g[0], my[6 * 64] = g[S.pool_cap], my[7 * 64] = g[2 * (size_t)S.pool_cap];
              my[8 * 64] = g[3 * (size_t)S.pool_cap], my[9 * 64] = g[4 * (size_t)S.pool_cap], my[10 * 64] = g[5 * (size_t)S.pool_cap];
              my[3 * 64] = g[6 * (size_t)S.pool_cap];
              meta[ln] = (uint32_t)__double_as_longlong(g[7 * (size_t)S.pool_cap]);
              have = true;
            }  // else: an entry that did not fit (pool_overflow is set, the host renders the frame again): nothing to fetch
          }
          pool_n -= k;
          mixed = true;
        }
      }
    }
    const unsigned long long c4 = region_clock<COUNT>();
    region_add<COUNT>(S, 7, c0, c1), region_add<COUNT>(S, 8, c1, c2), region_add<COUNT>(S, 9, c2, c3);
    region_add<COUNT>(S, 10, c3, c4);
  }
  rgb[0] = wsv[ln], rgb[1] = wsv[64 + ln], rgb[2] = wsv[128 + ln];
}

// LDS per workgroup: the triangle tile of the brute-force sweep (9 KiB) or the per-lane candidate queues of the
// accelerated sweep (CAND_CAP x BLOCK_ACCEL x 4 B), then one trace-state region per wave
constexpr size_t lds_doubles(bool accel, bool tris, bool refr) {
  return accel ? queue_doubles(tris) + ws_doubles(refr, accel ? tris : true) : (size_t)TILE_TRIS * 9 + (BLOCK_EXACT / 64) * ws_doubles(refr, accel ? tris : true);
}

struct RowMap {
  uint32_t block_rows, block_first, block_step, local_rows;
  uint32_t tile_row0;  // first tile row of this launch (a frame is launched in bands when its ray pools would be too big)
  uint32_t tiles_x, tiles_y;  // RTC_XCD_STRIP: the launch is a 1-D grid over the band's tiles, padded to a multiple of 8 strips
  // RTC_BUSY_FIRST: the tiles of this rectangle (band coordinates, [rx0, rx1) x [ry0, ry1)) are taken first, the others after
  // them; n_first = its tile count padded to whole strip groups (0: no rectangle)
  uint32_t rx0, ry0, rx1, ry1, n_first;
  // RTC_ROW_COST_ORDER: the rectangle's tile rows in the order they are taken (densest mesh parts first); n_order = 0: top-down
  uint32_t n_order;
  uint16_t row_order[272];
  // RTC_STRIP_ORDER: n_strips != 0: workgroup number w of the launch (in the order the XCDs take them: w % 8 is its XCD) does what item
  // strip_tab[w] says (strip_table below): bits 0-11 tile column, 12-24 tile row (band coordinates), 25-27 part: 0 the whole tile,
  // 1 / 2 its upper / lower 8 x 4 pixels, 3..6 one of its 4 x 4 quarters, 7: bits 0-24 number a group of sixteen tiles in the list
  // behind the table that are stored BLACK untraced; 0xffffffff: nothing (padding at the end of an XCD's list).  Every tile of the
  // band is rendered, or cleared, exactly once.
  const uint32_t* strip_tab;
  uint32_t n_strips;  // items
};

// One tile of pixels (tile_side x tile_side, at tile column bx / tile row by of this call's dense output).
template <bool REFR, bool ACCEL, bool COUNT, bool TRIS, bool POOLG>
__device__ __forceinline__ void render_tile(const DScene& S, const DCamera& C, const RowMap& R, int limit, float* out,
                                            double* out64, uint32_t bx, uint32_t by, double* s_tri, double* ws,
                                            uint32_t wave_id, uint32_t half) {
  // each wave owns an 8x8 quadrant of the 16x16 tile (compact ray bundles cull better than 16x4 strips)
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t tx = (lane & 7) + 8 * (wave & 1), ty = (lane >> 3) + 8 * (wave >> 1);
  const uint32_t px = bx * tile_side(ACCEL) + tx;
  const uint32_t ly = by * tile_side(ACCEL) + ty;  // row in this call's dense output
  // half != 0 (accelerated kernels of opaque scenes): this wave renders the upper / lower four rows of the tile, its other lanes idle
  const bool valid = px < C.hsize && ly < R.local_rows && (!ACCEL || half == 0u || (half < 3u ? (lane >> 5) == half - 1u : (((lane >> 2) & 1u) | ((lane >> 5) << 1)) == half - 3u));
  const uint32_t py = valid ? (R.block_first + (ly / R.block_rows) * R.block_step) * R.block_rows + ly % R.block_rows : 0u;
  if (ACCEL && RTC_BG_EXIT && S.n_top) {
    // Background waves.  Shape::intersect of a Group tests the WHOLE line against the group's box (shape.rs:399-425),
    // and World::intersect of a ray that fails every top-level box is empty, so the pixel is BLACK (world.rs:93).
    // Decided per wave with an f32 copy of ray_for_pixel: its direction is off by < 1e-6 relative, the boxes are
    // inflated by 1e-4 of the scene scale (100x that), and axes the reference's check_axis may treat as parallel
    // (|dir| < EPSILON, absolute) give no constraint.  Nothing of the f64 path has been computed yet.
    bool maybe = false;
    if (valid) {
      const float wx = (float)C.half_width - ((float)px + 0.5f) * (float)C.pixel_size;
      const float wy = (float)C.half_height - ((float)py + 0.5f) * (float)C.pixel_size;
      const float o3[3] = {(float)C.origin[0], (float)C.origin[1], (float)C.origin[2]};
      const float d3[3] = {(float)C.inv[0] * wx + (float)C.inv[1] * wy - (float)C.inv[2],
                           (float)C.inv[4] * wx + (float)C.inv[5] * wy - (float)C.inv[6],
                           (float)C.inv[8] * wx + (float)C.inv[9] * wy - (float)C.inv[10]};
      // the true direction is this vector normalised: scale the parallel threshold instead of the vector
      const float len = sqrtf(d3[0] * d3[0] + d3[1] * d3[1] + d3[2] * d3[2]);
      const float omax = fmaxf(fmaxf(fabsf(o3[0]), fabsf(o3[1])), fabsf(o3[2]));
      const float sum = fabsf(o3[0]) + fabsf(o3[1]) + fabsf(o3[2]) + len;
      if (!(sum < 1e30f) || !(len > 1e-30f)) maybe = true;  // NaN / huge / degenerate camera: no statement
      for (uint32_t g = 0; g < S.n_top && !maybe; ++g) {
        float ext = 0.f;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          if (S.top_mn[g][a] > -INFINITY) ext = fmaxf(ext, fabsf(S.top_mn[g][a]));
          if (S.top_mx[g][a] < INFINITY) ext = fmaxf(ext, fabsf(S.top_mx[g][a]));
        }
        const float mb = 1e-4f * (omax + ext);
        float tmin = -INFINITY, tmax = INFINITY;  // whole line
#pragma unroll
        for (int a = 0; a < 3; ++a) {
#if RTC_STRAIGHT_CLASS
          {
            const bool use = fabsf(d3[a]) > 4e-5f * len;
            const float inv = rcp_approx(d3[a]);
            const float t0 = (S.top_mn[g][a] - mb - o3[a]) * inv, t1 = (S.top_mx[g][a] + mb - o3[a]) * inv;
            tmin = use ? fmaxf(tmin, fminf(t0, t1)) : tmin;
            tmax = use ? fminf(tmax, fmaxf(t0, t1)) : tmax;
          }
#else
          if (fabsf(d3[a]) > 4e-5f * len) {  // |unit direction component| > 4e-5 > EPSILON: certainly not "parallel"
            const float inv = rcp_approx(d3[a]);
            const float t0 = (S.top_mn[g][a] - mb - o3[a]) * inv, t1 = (S.top_mx[g][a] + mb - o3[a]) * inv;
            tmin = fmaxf(tmin, fminf(t0, t1));
            tmax = fminf(tmax, fmaxf(t0, t1));
          }
#endif
        }
        if (!(tmin > tmax)) maybe = true;
      }
    }
    if (__ballot(maybe ? 1 : 0) == 0) {  // wave-uniform; accelerated waves never meet at a barrier
      if (COUNT) count_rays(S, 0, valid && limit >= 1), count_rays(S, 20, valid && limit >= 1);
      if (valid) {
        const size_t idx = ((size_t)ly * C.hsize + px) * 3;
        out[idx + 0] = 0.f, out[idx + 1] = 0.f, out[idx + 2] = 0.f;
        if (out64) out64[idx + 0] = 0.0, out64[idx + 1] = 0.0, out64[idx + 2] = 0.0;
      }
      return;
    }
  }
  timers_begin();
  const unsigned long long k0 = region_clock<COUNT>();
  Ray ray{{0, 0, 0}, {0, 0, 1}};
  if (valid) {
    // Camera::ray_for_pixel (camera.rs:48-65)
    double xoffset = ((double)px + 0.5) * C.pixel_size;
    double yoffset = ((double)py + 0.5) * C.pixel_size;
    double world_x = C.half_width - xoffset;
    double world_y = C.half_height - yoffset;
    const double* m = C.inv;
    // pixel = inv * point(world_x, world_y, -1): row i is m[4i]*wx + m[4i+1]*wy + m[4i+2]*-1.0 + m[4i+3]*1.0, summed left
    // to right; the last two products do not depend on the pixel (DCamera::zt, wt), nor does the origin
    V3 pixel = {m[0] * world_x + m[1] * world_y + C.zt[0] + C.wt[0], m[4] * world_x + m[5] * world_y + C.zt[1] + C.wt[1],
                m[8] * world_x + m[9] * world_y + C.zt[2] + C.wt[2]};
    V3 origin = {C.origin[0], C.origin[1], C.origin[2]};
    ray.o = origin;
    ray.d = normalize(pixel - origin);
  }
  double rgb[3];
  const unsigned long long k1 = region_clock<COUNT>();
  trace<REFR, ACCEL, COUNT, TRIS, POOLG>(S, ray, valid, limit, rgb, s_tri, ws, wave_id, ACCEL);  // (primary rays: one origin)
  const unsigned long long k2 = region_clock<COUNT>();
  // the pixel's coordinates are rebuilt from the thread index here (opaque to the optimiser), so that they do not
  // occupy registers -- or get spilled by every wave, background ones included -- during the whole trace
  uint32_t tid2 = threadIdx.x;
  asm volatile("" : "+v"(tid2));
  const uint32_t lane2 = tid2 & 63, wave2 = tid2 >> 6;
  const uint32_t px2 = bx * tile_side(ACCEL) + (lane2 & 7) + 8 * (wave2 & 1);
  const uint32_t ly2 = by * tile_side(ACCEL) + (lane2 >> 3) + 8 * (wave2 >> 1);
  if (px2 < C.hsize && ly2 < R.local_rows && (!ACCEL || half == 0u || (half < 3u ? (lane2 >> 5) == half - 1u : (((lane2 >> 2) & 1u) | ((lane2 >> 5) << 1)) == half - 3u))) {
    const size_t idx = ((size_t)ly2 * C.hsize + px2) * 3;  // Canvas layout: x + y*width (canvas.rs:20-26)
    out[idx + 0] = (float)rgb[0];
    out[idx + 1] = (float)rgb[1];
    out[idx + 2] = (float)rgb[2];
    if (out64) out64[idx + 0] = rgb[0], out64[idx + 1] = rgb[1], out64[idx + 2] = rgb[2];
  }
  const unsigned long long k3 = region_clock<COUNT>();
  region_add<COUNT>(S, 6, k0, k1), region_add<COUNT>(S, 11, k2, k3), region_add<COUNT>(S, 12, k0, k3);
  timers_end();
}

#if RTC_WAVE_TIMELINE
constexpr uint32_t TIMELINE_WGS = 1u << 18;
constexpr uint32_t TIMELINE_REC = 16;
// per workgroup: start, end, xcc | hw id << 8, bx | by << 16 | 1 << 40; with -DRTC_REGION_TIMERS=2 also the wave's own loop
// rounds and region cycles: [4..10] = counters 13..19, [11..15] = regions 7, 9, 22, 23, 24 (see the table above wave_count)
__device__ unsigned long long g_timeline[(size_t)TIMELINE_WGS * TIMELINE_REC];
#endif

template <bool REFR, bool ACCEL, bool COUNT, bool TRIS, bool POOLG>
__global__ __launch_bounds__(block_threads(ACCEL), ACCEL ? (TRIS ? (REFR ? RTC_MIN_WAVES : RTC_OPAQUE_MIN_WAVES) : RTC_PRIM_MIN_WAVES) : RTC_EXACT_MIN_WAVES) void render_kernel(DScene S, DCamera C, RowMap R, int limit, float* out,
                                                       double* out64) {
  __shared__ double s_tri[lds_doubles(ACCEL, TRIS, REFR)];
#if RTC_WAVE_TIMELINE
  const unsigned long long tl0 = __builtin_amdgcn_s_memrealtime();
#endif
  constexpr uint32_t WAVES = block_threads(ACCEL) / 64;
  double* ws = s_tri + (ACCEL ? queue_doubles(TRIS) : (size_t)TILE_TRIS * 9 + (threadIdx.x >> 6) * ws_doubles(REFR, ACCEL ? TRIS : true));
#if RTC_XCD_STRIP
  // Workgroups go to the 8 XCDs round-robin by their linear number; each XCD has its own L2.  Strips of RTC_XCD_STRIP consecutive
  // tiles are dealt to the XCDs instead of single tiles, so that neighbouring tiles (which meet the same clusters) share an L2.
  const uint32_t lin = blockIdx.x;
  const bool first = lin < R.n_first;
  const uint32_t l2 = first ? lin : lin - R.n_first, xcd = l2 & 7u, slot = l2 >> 3;
  uint32_t t = (slot / RTC_XCD_STRIP) * (8u * RTC_XCD_STRIP) + xcd * RTC_XCD_STRIP + (slot % RTC_XCD_STRIP);
  uint32_t bx_, by_;
  uint32_t half = 0;  // 0: the wave renders its whole tile; 1 / 2: its upper / lower 8 x 4 pixels; 3..6: one of its 4 x 4 quarters (strip_table)
  const uint32_t rw = R.rx1 - R.rx0, rh = R.ry1 - R.ry0;
  if (RTC_STRIP_ORDER && TRIS && R.n_strips) {  // (only scenes with triangle runs get a table: the other kernels stay without its registers)
    if (lin >= R.n_strips) return;
    const uint32_t it = R.strip_tab[lin];
    if (it == 0xffffffffu) return;  // padding
    const uint32_t part = (it >> 25) & 7u;
    if (part == 7u) {  // sixteen tiles that no leaf's bounds project onto (strip_table: RTC_CLEAR_EMPTY): BLACK
      const uint32_t n_clear = R.strip_tab[R.n_strips];  // (the list's length, then the list)
      const uint32_t* list = R.strip_tab + R.n_strips + 1;
      const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6, g = it & 0x1ffffffu;
      for (uint32_t k = g * 16u + wv; k < min(n_clear, g * 16u + 16u); k += block_threads(ACCEL) / 64) {
        const uint32_t t = list[k];
        const uint32_t px = (t & 0xfffu) * 8u + (lane & 7u), ly = (((t >> 12) & 0x1fffu) + R.tile_row0) * 8u + (lane >> 3);
        const bool valid = px < C.hsize && ly < R.local_rows;
        if (COUNT) count_rays(S, 0, valid && limit >= 1), count_rays(S, 20, valid && limit >= 1);
        if (valid) {
          const size_t idx = ((size_t)ly * C.hsize + px) * 3;
          out[idx + 0] = 0.f, out[idx + 1] = 0.f, out[idx + 2] = 0.f;
          if (out64) out64[idx + 0] = 0.0, out64[idx + 1] = 0.0, out64[idx + 2] = 0.0;
        }
      }
      return;
    }
    bx_ = it & 0xfffu, by_ = (it >> 12) & 0x1fffu, half = REFR ? 0u : part;
  } else if (first) {
    if (t >= rw * rh) return;  // padding
    const uint32_t k = t / rw;
    bx_ = R.rx0 + t % rw, by_ = R.ry0 + (R.n_order ? (uint32_t)R.row_order[k] : k);
  } else {
    const uint32_t above = R.ry0 * R.tiles_x, side_w = R.tiles_x - rw, mid = rh * side_w;
    if (t < above) {
      bx_ = t % R.tiles_x, by_ = t / R.tiles_x;
    } else if (t - above < mid) {
      t -= above;
      const uint32_t c = t % side_w;
      bx_ = c < R.rx0 ? c : c + rw, by_ = R.ry0 + t / side_w;
    } else {
      t -= above + mid;
      bx_ = t % R.tiles_x, by_ = R.ry1 + t / R.tiles_x;
    }
    if (by_ >= R.tiles_y) return;  // padding
  }
  const uint32_t wave_id = lin * WAVES + (WAVES > 1 ? (threadIdx.x >> 6) : 0u);
  render_tile<REFR, ACCEL, COUNT, TRIS, POOLG>(S, C, R, limit, out, out64, bx_, by_ + R.tile_row0, s_tri, ws, wave_id, (REFR || !TRIS) ? 0u : half);
#if RTC_WAVE_TIMELINE
  if (threadIdx.x == 0 && lin < TIMELINE_WGS) {
    uint32_t xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    unsigned long long* e = g_timeline + (size_t)lin * TIMELINE_REC;
    e[0] = tl0, e[1] = __builtin_amdgcn_s_memrealtime(), e[2] = (unsigned long long)(xcc & 0xfu) | ((unsigned long long)hw << 8);
    e[3] = (unsigned long long)bx_ | ((unsigned long long)by_ << 16) | (1ull << 40);
#if RTC_REGION_TIMERS == 2
    for (int k = 0; k < 7; ++k) e[4 + k] = s_timers[13 + k];
    e[11] = s_timers[7], e[12] = s_timers[9], e[13] = s_timers[22], e[14] = s_timers[23], e[15] = s_timers[24];
#endif
  }
#endif
#else
  const uint32_t wave_id = (blockIdx.y * gridDim.x + blockIdx.x) * WAVES + (WAVES > 1 ? (threadIdx.x >> 6) : 0u);  // scalar for one-wave workgroups
  // One workgroup per tile.  (A resident-wave variant that pulled tiles from an atomic counter was measured and
  // dropped: 32 400 returning atomics on one address serialise at ~13.6 ns each, a 0.44 ms floor per 1080p frame,
  // while dispatching 32 400 one-wave workgroups costs 11 us; profiles/r01/v10_*.)
  render_tile<REFR, ACCEL, COUNT, TRIS, POOLG>(S, C, R, limit, out, out64, blockIdx.x, blockIdx.y + R.tile_row0, s_tri, ws, wave_id, 0u);
#endif
}

template <bool REFR, bool ACCEL, bool COUNT, bool TRIS, bool POOLG>
__global__ __launch_bounds__(block_threads(ACCEL), ACCEL ? (TRIS ? (REFR ? RTC_MIN_WAVES : RTC_OPAQUE_MIN_WAVES) : RTC_PRIM_MIN_WAVES) : RTC_EXACT_MIN_WAVES) void color_at_kernel(DScene S, uint64_t n, const double* origins,
                                                         const double* dirs, int limit, float* out, double* out64) {
  __shared__ double s_tri[lds_doubles(ACCEL, TRIS, REFR)];
  constexpr uint32_t WAVES = block_threads(ACCEL) / 64;
  double* ws = s_tri + (ACCEL ? queue_doubles(TRIS) : (size_t)TILE_TRIS * 9 + (threadIdx.x >> 6) * ws_doubles(REFR, ACCEL ? TRIS : true));
  const uint32_t wave_id = blockIdx.x * WAVES + (WAVES > 1 ? (threadIdx.x >> 6) : 0u);
  const uint64_t i = (uint64_t)blockIdx.x * block_threads(ACCEL) + threadIdx.x;
  const bool valid = i < n;
  Ray ray{{0, 0, 0}, {0, 0, 1}};
  if (valid) {
    ray.o = {origins[3 * i], origins[3 * i + 1], origins[3 * i + 2]};
    ray.d = {dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]};
  }
  double rgb[3];
  trace<REFR, ACCEL, COUNT, TRIS, POOLG>(S, ray, valid, limit, rgb, s_tri, ws, wave_id);
  if (valid) {
    if (out) out[3 * i] = (float)rgb[0], out[3 * i + 1] = (float)rgb[1], out[3 * i + 2] = (float)rgb[2];
    if (out64) out64[3 * i] = rgb[0], out64[3 * i + 1] = rgb[1], out64[3 * i + 2] = rgb[2];
  }
}

__global__ void selftest_math_kernel(int op, uint64_t n, const double* a, const double* b, double* out) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (op == 5 || op == 6) {  // wave reductions used by the bundle construction: every lane of the wave takes part
    const float v = i < n ? (float)a[i] : (op == 5 ? INFINITY : -INFINITY);
    const float r = op == 5 ? wave_min(v) : wave_max(v);
    if (i < n) out[i] = (double)r;
    return;
  }
  if (i >= n) return;
  double x = a[i], y = b ? b[i] : 0.0, r;
  switch (op) {
    case 0: r = x / y; break;
    case 1: r = sqrt(x); break;
    case 2: r = pow(x, y); break;
    case 3: r = floor(x); break;
    default: r = fmod(x, y); break;
  }
  out[i] = r;
}

}  // namespace

// =====================================================================================================
//                                               host
// =====================================================================================================
namespace {

thread_local std::string g_err = "";
RtcStatus fail(RtcStatus st, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return st;
}
#define HIPCHECK(expr)                                                                                  \
  do {                                                                                                  \
    hipError_t e_ = (expr);                                                                             \
    if (e_ != hipSuccess) return fail(RTC_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace

// global ray-pool area per launch (bytes); RTC_POOL_BUDGET_MB in the environment overrides the default (tests use it to
// force the banded launches on small frames)
static uint64_t pool_budget() {
  if (const char* e = std::getenv("RTC_POOL_BUDGET_MB")) {
    const long long v = std::atoll(e);
    if (v > 0) return (uint64_t)v << 20;
  }
  return RTC_POOL_BUDGET_MB * (1ull << 20);
}
constexpr int RTC_N_COUNTERS = 32;  // [0..5] work counters (RtcStats), [6..19] wave-cycle region timers and loop rounds
                                    // (RTC_REGION_TIMERS), [20] primary rays answered by the background exit

struct RtcScene {
  int device = 0;
  bool accel_ok = false;
  DScene d{};
  std::vector<void*> allocs;
  bool recursive_material = false;  // any material with reflective != 0 or transparency != 0
  unsigned long long* d_counters = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool timed = false;
  hipStream_t last_stream = nullptr;
  double upload_ms = 0.0;
  // device buffers of rtc_color_at, kept between calls (grown on demand)
  double *d_ray_o = nullptr, *d_ray_d = nullptr, *d_ray_out64 = nullptr;
  float* d_ray_out32 = nullptr;
  size_t ray_cap = 0;
  // global part of the waves' ray pools (DScene::pool): one area per stream the scene has been launched on (launches on one
  // stream run in order and reuse it; launches on different streams may overlap and must not share one), sized per launch, kept
  struct PoolArea {
    hipStream_t stream;
    double* p;
    size_t bytes;
  };
  std::vector<PoolArea> pools;
  // world-space boxes (min xyz, max xyz) of the top-level groups: only a hint for the order in which a launch takes its tiles
  std::vector<std::array<double, 6>> top_boxes;
  // per mesh tile (<= 512 triangles, rtc_accel.hpp): its eight corners in world space and its triangle count -- where on the
  // canvas the many-small-triangle parts of a mesh are, for the order in which a launch takes its tile rows (launch_render)
  struct CostBox {
    double corner[8][3];
    double tris;
  };
  std::vector<CostBox> cost_boxes;
  // the same per cluster (<= 8 triangles), with the class of the run's materials (0 opaque, 1 reflective, 2 transparent): what
  // the tile order of a launch is estimated from (strip_table)
  struct BlockCost {
    double corner[8][3];
    float tris;
    int mat_class;
    uint32_t tri_first = 0, tri_count = 0;  // a cluster's triangles in tri_world (nine doubles each)
  };
  std::vector<BlockCost> cluster_costs;
  // world-space boxes of the bounded primitives (sphere, cube, truncated cylinder / cone), and whether any leaf has no bounds the
  // host can project (plane, untruncated cylinder / cone, geometry the accelerated path refuses): with the clusters' boxes, what
  // decides on which tiles NO leaf can be met (strip_table: RTC_CLEAR_EMPTY)
  std::vector<BlockCost> prim_bounds;
  bool unbounded_leaf = false;
  // world-space vertices (p1, p1 + e1, p1 + e2: nine doubles) of every triangle of every run, for scenes of at most TRI_WORLD_MAX
  // triangles: with them a tile counts as covered by a triangle run only where the pixel box of one of its TRIANGLES reaches, not
  // of one of its cluster boxes (RTC_TRI_COVER; 10-14 % of the tiles a C3 frame rendered held a cluster box and no triangle)
  std::vector<double> tri_world;
  // strip orders of recent launches, by everything they depend on (camera, rows, band, recursion limit): a scheduling hint
  // that is a pure function of those; at most STRIP_CACHE entries, oldest dropped
  struct StripTable {
    std::vector<unsigned char> key;
    uint32_t *host = nullptr, *dev = nullptr;  // host: pinned, what the copy on `uploaded_on` reads
    size_t cap = 0, n = 0;                  // entries the two buffers hold / the table has
    hipEvent_t copied = nullptr;            // recorded behind that copy: host may be rewritten once it has completed
    hipStream_t uploaded_on = nullptr;
    uint64_t used = 0;
  };
  std::vector<StripTable> strip_tables;
  uint64_t strip_clock = 0;
  // A camera that moves every frame never meets its table again, and building one costs the host four frames' worth of kernel time
  // (strip_table): after three tables in a row that no later launch used, a view gets its table only once it has been seen twice.
  uint32_t tables_unused_in_a_row = 0;
  std::array<uint64_t, 16> views_without_table{};  // hashes of the keys that went without (a ring)
  uint32_t views_without_table_at = 0;
  unsigned int* d_pool_flag = nullptr;
  // The global part of a wave's ray pool holds pool_scale x the entries the depth-first bound asks for (pool_entries()).  A
  // launch whose waves still overflow it reports that through d_pool_flag; check_pool_flag() then doubles the factor, and the
  // entry points that own the whole call (rtc_render, rtc_multi_render, rtc_color_at) render the frame again.
  uint32_t pool_scale = 1;
};
constexpr uint32_t POOL_SCALE_MAX = 64;

namespace {

template <class T>
RtcStatus upload(RtcScene* sc, const std::vector<T>& h, const T** out) {
  *out = nullptr;
  if (h.empty()) return RTC_OK;
  void* p = nullptr;
  HIPCHECK(hipMalloc(&p, h.size() * sizeof(T)));
  sc->allocs.push_back(p);
  HIPCHECK(hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
  *out = (const T*)p;
  return RTC_OK;
}

// Host mirror of tuple.rs normalize for the flatten-time triangle normals.
void h_normalize(double v[3]) {
  double m = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  if (m == 0.0) {
    v[0] = v[1] = v[2] = 0.0;
    return;
  }
  v[0] = v[0] / m, v[1] = v[1] / m, v[2] = v[2] / m;
}

// world-space corners of an object-space box under the object-from-world transform m = [A | t] (rows 0..2): p_world = A^-1 (p - t)
bool box_world_corners(const double* m, const double mn[3], const double mx[3], double corner[8][3]) {
  const double A[3][3] = {{m[0], m[1], m[2]}, {m[4], m[5], m[6]}, {m[8], m[9], m[10]}}, tr[3] = {m[3], m[7], m[11]};
  const double det = A[0][0] * (A[1][1] * A[2][2] - A[1][2] * A[2][1]) - A[0][1] * (A[1][0] * A[2][2] - A[1][2] * A[2][0]) +
                     A[0][2] * (A[1][0] * A[2][1] - A[1][1] * A[2][0]);
  if (!(std::fabs(det) > 1e-300)) return false;
  double I[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      I[j][i] = (A[(i + 1) % 3][(j + 1) % 3] * A[(i + 2) % 3][(j + 2) % 3] - A[(i + 1) % 3][(j + 2) % 3] * A[(i + 2) % 3][(j + 1) % 3]) / det;
  for (int k = 0; k < 8; ++k) {
    const double po[3] = {((k & 1) ? mx[0] : mn[0]) - tr[0], ((k & 2) ? mx[1] : mn[1]) - tr[1], ((k & 4) ? mx[2] : mn[2]) - tr[2]};
    for (int i = 0; i < 3; ++i) corner[k][i] = I[i][0] * po[0] + I[i][1] * po[1] + I[i][2] * po[2];
    for (int i = 0; i < 3; ++i)
      if (!std::isfinite(corner[k][i])) return false;
  }
  return true;
}

RtcStatus validate_limit(const RtcScene* sc, int limit) {
  if (limit < 0 || limit > RTC_MAX_RECURSION_LIMIT)
    return fail(RTC_ERR_INVALID_ARG, "recursion_limit %d outside [0, %d]", limit, RTC_MAX_RECURSION_LIMIT);
  if (limit % 3 == 1 && sc->recursive_material)
    return fail(RTC_ERR_DEPTH_UNDERFLOW,
                "recursion_limit %d = 1 (mod 3) with a reflective/transparent material: the reference underflows "
                "`remaining - 1` (world.rs:68)",
                limit);
  return RTC_OK;
}

}  // namespace

extern "C" {

int rtc_abi_version(void) { return RTC_ABI_VERSION; }

int rtc_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

const char* rtc_last_error_string(void) { return g_err.c_str(); }

uint32_t rtc_local_rows(uint32_t vsize, const RtcRenderOpts* opts) {
  uint32_t B = (opts && opts->block_rows) ? opts->block_rows : vsize;
  uint32_t first = (opts && opts->block_rows) ? opts->block_first : 0;
  uint32_t step = (opts && opts->block_rows && opts->block_step) ? opts->block_step : 1;
  if (B == 0) return 0;
  uint32_t nblocks = (vsize + B - 1) / B;
  uint32_t limit = (opts && opts->block_rows && opts->block_count) ? opts->block_count : 0xffffffffu;
  uint32_t rows = 0;
  for (uint32_t bi = first; bi < nblocks && limit; bi += step, --limit) rows += (bi * B + B <= vsize) ? B : (vsize - bi * B);
  return rows;
}

RtcStatus rtc_scene_create(const RtcSceneDesc* desc, int device, RtcScene** out) {
  if (!desc || !out) return fail(RTC_ERR_INVALID_ARG, "null argument");
  *out = nullptr;
  int ndev = rtc_device_count();
  if (ndev <= 0) return fail(RTC_ERR_NO_DEVICE, "no HIP device visible");
  if (device < 0 || device >= ndev) return fail(RTC_ERR_NO_DEVICE, "device %d out of range (%d devices)", device, ndev);
  const RtcSceneDesc& D = *desc;
  if (D.n_leaves && (!D.leaf_kind || !D.leaf_transform || !D.leaf_material || !D.leaf_group))
    return fail(RTC_ERR_INVALID_ARG, "null leaf array");
  if ((D.n_transforms && !D.transform_inverse) || (D.n_materials && !D.materials) || (D.n_patterns && !D.patterns) ||
      (D.n_groups && (!D.group_parent || !D.group_bounds_min || !D.group_bounds_max)))
    return fail(RTC_ERR_INVALID_ARG, "null table");

  double t0 = now_ms();
  // ---- transforms: keep rows 0..2; the bottom row must be exactly (0,0,0,1) (affine) -------------
  std::vector<double> xf((size_t)D.n_transforms * 12);
  for (uint32_t i = 0; i < D.n_transforms; ++i) {
    const double* m = D.transform_inverse + (size_t)i * 16;
    if (!(m[12] == 0.0 && m[13] == 0.0 && m[14] == 0.0 && m[15] == 1.0))
      return fail(RTC_ERR_NON_AFFINE, "transform_inverse %u has bottom row (%g,%g,%g,%g)", i, m[12], m[13], m[14], m[15]);
    std::memcpy(&xf[(size_t)i * 12], m, 12 * sizeof(double));
  }
  // ---- materials / patterns ----------------------------------------------------------------------
  std::vector<DMaterial> mats(D.n_materials);
  bool has_refr = false, recursive = false;
  for (uint32_t i = 0; i < D.n_materials; ++i) {
    const RtcMaterial& s = D.materials[i];
    DMaterial& m = mats[i];
    std::memcpy(m.color, s.color, sizeof m.color);
    m.ambient = s.ambient, m.diffuse = s.diffuse, m.specular = s.specular, m.shininess = s.shininess;
    m.reflective = s.reflective, m.transparency = s.transparency, m.refractive_index = s.refractive_index;
    m.pattern = s.pattern, m._pad = 0;
    if (s.pattern >= (int32_t)D.n_patterns) return fail(RTC_ERR_INVALID_ARG, "material %u: pattern %d out of range", i, s.pattern);
    if (s.transparency != 0.0) has_refr = true;
    if (s.transparency != 0.0 || s.reflective != 0.0) recursive = true;
  }
  std::vector<DPattern> pats(D.n_patterns);
  for (uint32_t i = 0; i < D.n_patterns; ++i) {
    const RtcPattern& s = D.patterns[i];
    if (s.kind < RTC_PATTERN_STRIPE || s.kind > RTC_PATTERN_TEST) return fail(RTC_ERR_INVALID_ARG, "pattern %u: kind %d", i, s.kind);
    const double* m = s.transform_inverse;
    if (!(m[12] == 0.0 && m[13] == 0.0 && m[14] == 0.0 && m[15] == 1.0))
      return fail(RTC_ERR_NON_AFFINE, "pattern %u transform_inverse is not affine", i);
    DPattern& p = pats[i];
    p.kind = s.kind, p._pad = 0;
    std::memcpy(p.a, s.a, sizeof p.a);
    std::memcpy(p.b, s.b, sizeof p.b);
    std::memcpy(p.inv, m, 12 * sizeof(double));
  }
  // ---- groups ------------------------------------------------------------------------------------
  std::vector<DGroup> groups(D.n_groups);
  for (uint32_t g = 0; g < D.n_groups; ++g) {
    if (D.group_parent[g] >= (int32_t)g) return fail(RTC_ERR_INVALID_ARG, "group %u: parent %d is not an earlier group", g, D.group_parent[g]);
    for (int k = 0; k < 3; ++k) groups[g].mn[k] = D.group_bounds_min[3 * g + k], groups[g].mx[k] = D.group_bounds_max[3 * g + k];
    groups[g].parent = D.group_parent[g] < 0 ? -1 : D.group_parent[g];
    groups[g]._pad = 0;
    for (int k = 0; k < 3; ++k) {
      groups[g].fmn[k] = std::isfinite(groups[g].mn[k]) ? f32_down(groups[g].mn[k]) : -INFINITY;
      groups[g].fmx[k] = std::isfinite(groups[g].mx[k]) ? f32_up(groups[g].mx[k]) : INFINITY;
    }
    double ext = 0.0;
    for (int k = 0; k < 3; ++k) {
      if (std::isfinite(groups[g].mn[k])) ext = std::fmax(ext, std::fabs(groups[g].mn[k]));
      if (std::isfinite(groups[g].mx[k])) ext = std::fmax(ext, std::fabs(groups[g].mx[k]));
    }
    groups[g].fext = f32_up(ext);
    groups[g]._pad2 = 0.f;
  }
  // ---- leaves -> segments + packed triangles -----------------------------------------------------
  std::vector<uint8_t> kind(D.n_leaves);
  std::vector<uint32_t> lxf(D.n_leaves), lmat(D.n_leaves);
  std::vector<double> params((size_t)D.n_leaves * 3, 0.0);
  std::vector<DSegment> segs;
  std::vector<double> tri9, leaf_nw((size_t)D.n_leaves * 3, 0.0);
  uint32_t n_tri = 0;
  for (uint32_t l = 0; l < D.n_leaves; ++l) {
    uint8_t k = D.leaf_kind[l];
    if (k > RTC_TRIANGLE) return fail(RTC_ERR_INVALID_ARG, "leaf %u: kind %u", l, k);
    if (D.leaf_transform[l] >= D.n_transforms) return fail(RTC_ERR_INVALID_ARG, "leaf %u: transform %u out of range", l, D.leaf_transform[l]);
    if (D.leaf_material[l] >= D.n_materials) return fail(RTC_ERR_INVALID_ARG, "leaf %u: material %u out of range", l, D.leaf_material[l]);
    if (D.leaf_group[l] >= (int32_t)D.n_groups) return fail(RTC_ERR_INVALID_ARG, "leaf %u: group %d out of range", l, D.leaf_group[l]);
    if (D.leaf_class && D.leaf_class[l] > l) return fail(RTC_ERR_INVALID_ARG, "leaf %u: class %u is not an earlier leaf", l, D.leaf_class[l]);
    kind[l] = k;
    lxf[l] = D.leaf_transform[l];
    lmat[l] = D.leaf_material[l];
    const int32_t grp = D.leaf_group[l] < 0 ? -1 : D.leaf_group[l];
    if ((k == RTC_CYLINDER || k == RTC_CONE)) {
      if (!D.leaf_params) return fail(RTC_ERR_INVALID_ARG, "leaf %u: leaf_params is null", l);
      for (int j = 0; j < 3; ++j) params[(size_t)l * 3 + j] = D.leaf_params[(size_t)l * 3 + j];
    }
    if (k == RTC_TRIANGLE) {
      if (!D.leaf_triangle || !D.tri_p1 || !D.tri_e1 || !D.tri_e2 || !D.tri_normal) return fail(RTC_ERR_INVALID_ARG, "null triangle array");
      uint32_t ti = D.leaf_triangle[l];
      if (ti >= D.n_triangles) return fail(RTC_ERR_INVALID_ARG, "leaf %u: triangle %u out of range", l, ti);
      for (int j = 0; j < 3; ++j) tri9.push_back(D.tri_p1[(size_t)ti * 3 + j]);
      for (int j = 0; j < 3; ++j) tri9.push_back(D.tri_e1[(size_t)ti * 3 + j]);
      for (int j = 0; j < 3; ++j) tri9.push_back(D.tri_e2[(size_t)ti * 3 + j]);
      // Shape::normal_at for a triangle does not depend on the point (shape.rs:509): evaluate it once here,
      // with the same two normalisations (normal_to_world :623-635, then normal_at :513-518).
      const double* inv = &xf[(size_t)lxf[l] * 12];
      const double* n = D.tri_normal + (size_t)ti * 3;
      double wn[3] = {inv[0] * n[0] + inv[4] * n[1] + inv[8] * n[2], inv[1] * n[0] + inv[5] * n[1] + inv[9] * n[2],
                      inv[2] * n[0] + inv[6] * n[1] + inv[10] * n[2]};
      h_normalize(wn);
      h_normalize(wn);
      leaf_nw[(size_t)l * 3] = wn[0], leaf_nw[(size_t)l * 3 + 1] = wn[1], leaf_nw[(size_t)l * 3 + 2] = wn[2];
      if (!segs.empty() && segs.back().is_tri && segs.back().xform == lxf[l] && segs.back().group == grp &&
          segs.back().first_leaf + segs.back().count == l) {
        segs.back().count++;
      } else {
        segs.push_back(DSegment{1u, l, 1u, lxf[l], grp, n_tri, 0u, 0u, 0u, 0u, 0.f, 0.f, 0u, 0u, 0u, (uint32_t)K_TRIANGLE});
      }
      ++n_tri;
    } else {
      segs.push_back(DSegment{0u, l, 1u, lxf[l], grp, 0u, 0u, 0u, 0u, 0u, 0.f, 0.f, 0u, 0u, 0u, (uint32_t)k});
    }
  }
  // ---- group chains: for every segment the groups above it, root first, and the prefix shared with the previous
  //      segment (GroupGate in rtc_device.hpp evaluates only the rest) ---------------------------------------
  std::vector<uint32_t> seg_chain;
  {
    std::vector<uint32_t> prev, cur;
    for (DSegment& sg : segs) {
      cur.clear();
      for (int32_t g = sg.group; g >= 0; g = groups[(size_t)g].parent) cur.push_back((uint32_t)g);
      std::reverse(cur.begin(), cur.end());
      if (cur.size() > 64)
        return fail(RTC_ERR_TOO_MANY_GROUPS, "leaf %u sits %zu groups deep (limit %d)", sg.first_leaf, cur.size(), RTC_MAX_GROUP_DEPTH);
      uint32_t common = 0;
      while (common < cur.size() && common < prev.size() && cur[common] == prev[common]) ++common;
      sg.chain_off = (uint32_t)seg_chain.size();
      sg.chain_len = (uint32_t)cur.size();
      sg.chain_common = common;
      seg_chain.insert(seg_chain.end(), cur.begin(), cur.end());
      prev = cur;
    }
  }
  // ---- value-equal duplicate leaves (only matter for the containers bookkeeping of transparent scenes) -------
  std::vector<uint8_t> cflags;
  std::vector<uint32_t> lclass;
  if (has_refr && D.leaf_class) {
    bool dup = false;
    for (uint32_t l = 0; l < D.n_leaves; ++l) dup = dup || D.leaf_class[l] != l;
    if (dup) {
      std::vector<uint32_t> size(D.n_leaves, 0), last(D.n_leaves, 0);
      for (uint32_t l = 0; l < D.n_leaves; ++l) {
        const uint32_t c = D.leaf_class[D.leaf_class[l]] == D.leaf_class[l] ? D.leaf_class[l] : l;  // tolerate non-canonical input
        size[c]++, last[c] = l;
      }
      cflags.assign(D.n_leaves, 0);
      lclass.assign(D.n_leaves, 0);
      for (uint32_t l = 0; l < D.n_leaves; ++l) {
        const uint32_t c = D.leaf_class[D.leaf_class[l]] == D.leaf_class[l] ? D.leaf_class[l] : l;
        lclass[l] = c;
        cflags[l] = (uint8_t)((last[c] == l ? 1 : 0) | ((size[c] & 1u) ? 2 : 0));
      }
    }
  }
  // ---- acceleration data per triangle run (rtc_accel.hpp): spatial order, boxes, f32 filter records ------
  std::vector<DTriF> trif;
  std::vector<DBox> clusters, tiles, blocks;
  std::vector<RtcScene::CostBox> cost_boxes;
  std::vector<RtcScene::BlockCost> cluster_costs;
  std::vector<double> tri_world;    // RtcScene::tri_world
  bool tri_world_ok = RTC_TRI_COVER && n_tri <= TRI_WORLD_MAX;
  std::vector<double> tri9x;        // exact records, same (padded, spatial) indexing as trif
  bool accel_ok = true;
  for (double v : tri9)
    if (!(std::fabs(v) <= 1e30)) accel_ok = false;  // non-finite or huge geometry: no usable f32 boxes, exact sweep
  for (double v : xf)
    if (!std::isfinite(v)) accel_ok = false;
  if (accel_ok) {
    for (DSegment& sg : segs) {
      if (!sg.is_tri) continue;
      AccelRun run = build_accel_run(&tri9[(size_t)sg.first_tri * 9], sg.count);
      const uint32_t rec_base = (uint32_t)trif.size();  // records are padded: their own index space
      const uint32_t cl_base = (uint32_t)clusters.size();
      sg.first_tile = (uint32_t)tiles.size();
      sg.n_tiles = (uint32_t)run.tiles.size();
      sg.first_cluster = cl_base;
      sg.kind = (uint32_t)K_TRIANGLE | ((uint32_t)run.root_axis << 8);
      sg.first_rec = rec_base;
      for (size_t c = 0; c < run.clusters.size(); ++c) {  // the layout the triangle-level bundle test relies on
        const bool last = c + 1 == run.clusters.size();
        if (run.clusters[c].first != c * (size_t)CLUSTER_TRIS || (!last && run.clusters[c].count != (uint32_t)CLUSTER_TRIS))
          return fail(RTC_ERR_INVALID_ARG, "internal: cluster %zu of a triangle run is not densely packed", c);
      }
      sg.pmax = run.pmax;
      sg.ext = run.ext;
      for (size_t k = 0; k < run.order.size(); ++k) {
        if (run.order[k] == NO_TRIANGLE) {  // padding record: all zeros, never a hit
          tri9x.insert(tri9x.end(), TRI9X_STRIDE, 0.0);
        } else {
          const double* src = &tri9[(size_t)(sg.first_tri + run.order[k]) * 9];
          tri9x.insert(tri9x.end(), src, src + 9);
          {  // the record's tenth value: the depth-first leaf index, so that an exact test fetches one contiguous 80-byte record
            const unsigned long long leaf_bits = sg.first_leaf + run.order[k];
            double as_double;
            std::memcpy(&as_double, &leaf_bits, sizeof as_double);
            tri9x.push_back(as_double);
          }
        }
      }
      trif.insert(trif.end(), run.trif.begin(), run.trif.end());
      for (DBox b : run.clusters) {
        b.first += rec_base;
        clusters.push_back(b);
      }
      // world-space corners of an object-space box: p_world = A^-1 (p_object - t) for object-from-world [A | t]
      double I[3][3], tr[3];
      bool have_inv = false;
      {
        const double* m = &xf[(size_t)sg.xform * 12];
        const double A[3][3] = {{m[0], m[1], m[2]}, {m[4], m[5], m[6]}, {m[8], m[9], m[10]}};
        tr[0] = m[3], tr[1] = m[7], tr[2] = m[11];
        const double det = A[0][0] * (A[1][1] * A[2][2] - A[1][2] * A[2][1]) - A[0][1] * (A[1][0] * A[2][2] - A[1][2] * A[2][0]) +
                           A[0][2] * (A[1][0] * A[2][1] - A[1][1] * A[2][0]);
        if (std::fabs(det) > 1e-300) {
          have_inv = true;
          for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j)
              I[j][i] = (A[(i + 1) % 3][(j + 1) % 3] * A[(i + 2) % 3][(j + 2) % 3] - A[(i + 1) % 3][(j + 2) % 3] * A[(i + 2) % 3][(j + 1) % 3]) / det;
        }
      }
      auto world_corners = [&](const DBox& b, double corner[8][3]) {
        for (int k = 0; k < 8; ++k) {
          const double po[3] = {((k & 1) ? b.mx[0] : b.mn[0]) - tr[0], ((k & 2) ? b.mx[1] : b.mn[1]) - tr[1], ((k & 4) ? b.mx[2] : b.mn[2]) - tr[2]};
          for (int i = 0; i < 3; ++i) corner[k][i] = I[i][0] * po[0] + I[i][1] * po[1] + I[i][2] * po[2];
        }
      };
      int run_class = 0;  // what a hit on this run can spawn: scales the estimate of a wave's life (strip_table)
      for (uint32_t l = sg.first_leaf; l < sg.first_leaf + sg.count; ++l) {
        const RtcMaterial& mt = D.materials[lmat[l]];
        run_class = std::max(run_class, mt.transparency != 0.0 ? 2 : (mt.reflective != 0.0 ? 1 : 0));
      }
      if (!have_inv) tri_world_ok = false;
      if (have_inv)
        for (const DBox& cb : run.clusters) {
          RtcScene::BlockCost bc;
          bc.tris = (float)std::min<uint32_t>(cb.count, sg.count - std::min(sg.count, cb.first)), bc.mat_class = run_class;
          world_corners(cb, bc.corner);
          if (tri_world_ok) {  // the cluster's triangles, in the order of its records (padding records are no triangles)
            bc.tri_first = (uint32_t)(tri_world.size() / 9);
            for (uint32_t k = cb.first; k < cb.first + cb.count && k < run.order.size(); ++k) {
              if (run.order[k] == NO_TRIANGLE) continue;
              const double* q = &tri9[(size_t)(sg.first_tri + run.order[k]) * 9];
              for (int v = 0; v < 3; ++v) {  // p1, p1 + e1, p1 + e2: the corners of what Shape::intersect's Triangle arm tests (shape.rs:438-459)
                const double po[3] = {q[0] + (v == 1 ? q[3] : v == 2 ? q[6] : 0.0) - tr[0], q[1] + (v == 1 ? q[4] : v == 2 ? q[7] : 0.0) - tr[1],
                                      q[2] + (v == 1 ? q[5] : v == 2 ? q[8] : 0.0) - tr[2]};
                for (int i = 0; i < 3; ++i) tri_world.push_back(I[i][0] * po[0] + I[i][1] * po[1] + I[i][2] * po[2]);
              }
            }
            bc.tri_count = (uint32_t)(tri_world.size() / 9) - bc.tri_first;
          }
          cluster_costs.push_back(bc);
        }
      for (DBox b : run.tiles) {
        for (uint32_t k = 0; k < 8; ++k) {  // eight aligned runs of eight clusters: the middle level for per-ray tests
          DBox blk{{INFINITY, INFINITY, INFINITY}, 0u, {-INFINITY, -INFINITY, -INFINITY}, 0u};
          for (uint32_t c = b.first + 8 * k; c < b.first + b.count && c < b.first + 8 * (k + 1); ++c)
            for (int a = 0; a < 3; ++a)
              blk.mn[a] = std::fmin(blk.mn[a], run.clusters[c].mn[a]), blk.mx[a] = std::fmax(blk.mx[a], run.clusters[c].mx[a]);
          blocks.push_back(blk);
        }
        if (have_inv) {
          RtcScene::CostBox cb;
          cb.tris = (double)b.count * CLUSTER_TRIS;
          world_corners(b, cb.corner);
          cost_boxes.push_back(cb);
        }
        b.first += cl_base;
        tiles.push_back(b);
      }
    }
  }

  // Bounds of the leaves that are not in triangle runs (RtcScene::prim_bounds).  A sphere / cube fills [-1, 1]^3 of its object
  // space, a cylinder x, z in [-1, 1] between its (finite) truncation planes, or up to the radius of its caps if that is larger.
  std::vector<RtcScene::BlockCost> prim_bounds;
  bool unbounded_leaf = !accel_ok;
  for (const DSegment& sg : segs) {
    if (sg.is_tri) {
      if (sg.count && cluster_costs.empty()) unbounded_leaf = true;  // (a run without projectable cluster boxes)
      continue;
    }
    const uint32_t l = sg.first_leaf, k = sg.kind & 0xffu;
    double mn[3] = {-1.0, -1.0, -1.0}, mx[3] = {1.0, 1.0, 1.0};
    // A plane has no bounds; neither has a cone in the reference: a ray (nearly) parallel to one of its halves (|a| < EPSILON) gets the
    // root -c / 2b WITHOUT the test against minimum / maximum (shape.rs:368-373), anywhere on the infinite double cone.
    if (k == K_PLANE || k == K_CONE) { unbounded_leaf = true; continue; }
    if (k == K_CYLINDER) {
      const double lo = params[(size_t)l * 3], hi = params[(size_t)l * 3 + 1];
      if (!std::isfinite(lo) || !std::isfinite(hi)) { unbounded_leaf = true; continue; }
      mn[1] = std::fmin(lo, hi), mx[1] = std::fmax(lo, hi);
      double rr = 1.0;
      // a cap is the disc x^2 + z^2 <= |y| in the reference (check_cap, shape.rs:579-585), i.e. of radius sqrt(|y|): it sticks out of
      // a cylinder whose cap is not at |y| = 1 (the quirk is kept: SURVEY.md section 8a row 5 (ii))
      if (params[(size_t)l * 3 + 2] != 0.0) rr = std::fmax(rr, std::fmax(std::sqrt(std::fabs(lo)), std::sqrt(std::fabs(hi))));
      mn[0] = mn[2] = -rr, mx[0] = mx[2] = rr;
    }
    for (int a = 0; a < 3; ++a) mn[a] -= 1e-6 * (1.0 + std::fabs(mn[a])), mx[a] += 1e-6 * (1.0 + std::fabs(mx[a]));
    RtcScene::BlockCost bc;
    bc.tris = 0.f, bc.mat_class = 0;
    if (box_world_corners(&xf[(size_t)sg.xform * 12], mn, mx, bc.corner)) prim_bounds.push_back(bc);
    else unbounded_leaf = true;
  }
  {  // every triangle run must have produced its cluster boxes (a singular transform does not)
    size_t run_clusters = 0;
    for (const DSegment& sg : segs)
      if (sg.is_tri) run_clusters += (sg.count + CLUSTER_TRIS - 1) / CLUSTER_TRIS;
    if (cluster_costs.size() < run_clusters) unbounded_leaf = true;
  }

  auto* sc = new RtcScene;
  sc->device = device;
  sc->recursive_material = recursive;
  RtcStatus st = RTC_OK;
  auto bail = [&](RtcStatus s) {
    rtc_scene_destroy(sc);
    return s;
  };
  if (hipSetDevice(device) != hipSuccess) return bail(fail(RTC_ERR_HIP, "hipSetDevice(%d) failed", device));
  DScene& d = sc->d;
  std::memcpy(d.light_pos, D.light_position, sizeof d.light_pos);
  std::memcpy(d.light_int, D.light_intensity, sizeof d.light_int);
  if ((st = upload(sc, xf, &d.xf)) != RTC_OK) return bail(st);
  if ((st = upload(sc, mats, &d.mats)) != RTC_OK) return bail(st);
  if ((st = upload(sc, pats, &d.pats)) != RTC_OK) return bail(st);
  if ((st = upload(sc, groups, &d.groups)) != RTC_OK) return bail(st);
  if ((st = upload(sc, seg_chain, &d.seg_chain)) != RTC_OK) return bail(st);
  std::vector<DSegRec> segrecs(segs.size());
  for (size_t i = 0; i < segs.size(); ++i) {  // after the acceleration build: it fills in fields of the segments
    DSegRec& rc = segrecs[i];
    std::memset(&rc, 0, sizeof rc);
    rc.s = segs[i];
    std::memcpy(rc.xf, &xf[(size_t)segs[i].xform * 12], sizeof rc.xf);
    for (uint32_t dd = 0; dd < segs[i].chain_len && dd < (uint32_t)SEG_INLINE_GROUPS; ++dd)
      rc.g[dd] = groups[seg_chain[segs[i].chain_off + dd]];
  }
  if ((st = upload(sc, segrecs, &d.segs)) != RTC_OK) return bail(st);
  if ((st = upload(sc, kind, &d.leaf_kind)) != RTC_OK) return bail(st);
  if ((st = upload(sc, lxf, &d.leaf_xf)) != RTC_OK) return bail(st);
  if ((st = upload(sc, lmat, &d.leaf_mat)) != RTC_OK) return bail(st);
  if ((st = upload(sc, params, &d.leaf_params)) != RTC_OK) return bail(st);
  if ((st = upload(sc, tri9, &d.tri9)) != RTC_OK) return bail(st);
  if ((st = upload(sc, leaf_nw, &d.leaf_nw)) != RTC_OK) return bail(st);
  if ((st = upload(sc, cflags, &d.leaf_cflags)) != RTC_OK) return bail(st);
  if ((st = upload(sc, lclass, &d.leaf_class)) != RTC_OK) return bail(st);
  sc->accel_ok = accel_ok;
  if (accel_ok) {
    if ((st = upload(sc, trif, &d.trif)) != RTC_OK) return bail(st);
    if ((st = upload(sc, clusters, &d.clusters)) != RTC_OK) return bail(st);
    if ((st = upload(sc, tiles, &d.tiles)) != RTC_OK) return bail(st);
    if ((st = upload(sc, blocks, &d.blocks)) != RTC_OK) return bail(st);
    if ((st = upload(sc, tri9x, &d.tri9x)) != RTC_OK) return bail(st);
  }
  d.n_groups = D.n_groups, d.n_segs = (uint32_t)segs.size(), d.n_leaves = D.n_leaves, d.n_tri = n_tri;
  sc->cost_boxes = std::move(cost_boxes);
  sc->cluster_costs = std::move(cluster_costs);
  sc->prim_bounds = std::move(prim_bounds);
  if (tri_world_ok) sc->tri_world = std::move(tri_world);
  sc->unbounded_leaf = unbounded_leaf;
  for (uint32_t g = 0; g < D.n_groups; ++g)
    if (groups[g].parent < 0 && groups[g].fext < 1e30f)
      sc->top_boxes.push_back({(double)groups[g].fmn[0], (double)groups[g].fmn[1], (double)groups[g].fmn[2], (double)groups[g].fmx[0],
                               (double)groups[g].fmx[1], (double)groups[g].fmx[2]});
  {  // background early-out data: all leaves under <= MAX_TOP top-level groups with usable boxes
    d.n_top = 0;
    bool ok = D.n_leaves > 0;
    for (uint32_t l = 0; l < D.n_leaves && ok; ++l) ok = D.leaf_group[l] >= 0;
    std::vector<uint32_t> tops;
    for (uint32_t g = 0; g < D.n_groups && ok; ++g)
      if (groups[g].parent < 0) {
        tops.push_back(g);
        if (tops.size() > (size_t)MAX_TOP || !(groups[g].fext < 1e30f)) ok = false;
      }
    if (ok && !tops.empty()) {
      d.n_top = (uint32_t)tops.size();
      for (size_t i = 0; i < tops.size(); ++i)
        for (int a = 0; a < 3; ++a) d.top_mn[i][a] = groups[tops[i]].fmn[a], d.top_mx[i][a] = groups[tops[i]].fmx[a];
    }
  }
  d.has_refraction = has_refr ? 1u : 0u;
  d.counters = nullptr;
  {
    void* p = nullptr;
    if (hipMalloc(&p, RTC_N_COUNTERS * sizeof(unsigned long long)) != hipSuccess) return bail(fail(RTC_ERR_HIP, "hipMalloc(counters) failed"));
    sc->allocs.push_back(p);
    sc->d_counters = (unsigned long long*)p;
    if (hipMemset(p, 0, RTC_N_COUNTERS * sizeof(unsigned long long)) != hipSuccess) return bail(fail(RTC_ERR_HIP, "hipMemset(counters) failed"));
  }
  {
    void* p = nullptr;
    if (hipMalloc(&p, sizeof(unsigned int)) != hipSuccess || hipMemset(p, 0, sizeof(unsigned int)) != hipSuccess)
      return bail(fail(RTC_ERR_HIP, "hipMalloc(pool flag) failed"));
    sc->allocs.push_back(p);
    sc->d_pool_flag = (unsigned int*)p;
    d.pool = nullptr, d.pool_cap = 0, d.pool_overflow = sc->d_pool_flag;
  }
  if (hipEventCreate(&sc->ev0) != hipSuccess || hipEventCreate(&sc->ev1) != hipSuccess)
    return bail(fail(RTC_ERR_HIP, "hipEventCreate failed"));
  sc->upload_ms = now_ms() - t0;
  *out = sc;
  return RTC_OK;
}

void rtc_scene_destroy(RtcScene* sc) {
  if (!sc) return;
  (void)hipSetDevice(sc->device);
  for (void* p : sc->allocs) (void)hipFree(p);
  (void)hipFree(sc->d_ray_o), (void)hipFree(sc->d_ray_d), (void)hipFree(sc->d_ray_out64), (void)hipFree(sc->d_ray_out32);
  for (auto& a : sc->pools) (void)hipFree(a.p);
  for (auto& t : sc->strip_tables) {
    (void)hipFree(t.dev), (void)hipHostFree(t.host);
    if (t.copied) (void)hipEventDestroy(t.copied);
  }
  if (sc->ev0) (void)hipEventDestroy(sc->ev0);
  if (sc->ev1) (void)hipEventDestroy(sc->ev1);
  delete sc;
}

// The global part of the ray pools of a launch with n_waves waves (trace()).  With the depth-first walk of one pixel a wave
// has at most 64 * (generations - 1) rays parked at once, generations = ceil(L / 3): one ray per level and pixel (for two
// generations that is immediate: the 64 primaries have at most 64 second children, which have none).  The first WS_POOL_LDS
// of them live in LDS; the buffer holds the rest, plus 64 spare entries per wave from three generations on.  From three
// generations on the bound is an argument about the refill order, not a proof: rays of one pixel parked by different lanes
// can be in flight at once after a batch refill.  So the area is sized pool_scale x that figure, a wave that would exceed it
// sets pool_overflow instead of storing (nothing is corrupted, the frame is incomplete), and the host enlarges the pools and
// renders again (check_pool_flag).
static uint32_t pool_entries(const RtcScene* sc, int L) {  // global entries per wave; 0: the kernel variant without the overflow path
  const uint32_t in_lds = (uint32_t)WS_POOL_LDS;
  const int G = (L + 2) / 3;
  if (!sc->d.has_refraction || G < 2) return 0;
  const uint32_t bound = 64u * (uint32_t)(G - 1);
  uint32_t base = (bound > in_lds ? bound - in_lds : 0u) + (G >= 3 ? 64u : 0u);
  if (base)  // tests shrink the area to see the overflow, and the second render, happen
    if (const char* e = std::getenv("RTC_POOL_BASE_ENTRIES")) base = (uint32_t)std::max(1, std::atoi(e));
  return base * sc->pool_scale;
}
static RtcStatus ensure_pool(RtcScene* sc, int L, uint64_t n_waves, bool kernel_tris, DScene& S, hipStream_t stream) {
  (void)kernel_tris;
  const uint32_t cap = pool_entries(sc, L);
  const size_t need = (size_t)n_waves * cap * 64;
  S.pool = nullptr, S.pool_cap = cap, S.pool_overflow = sc->d_pool_flag;
  if (need == 0) return RTC_OK;
  RtcScene::PoolArea* area = nullptr;
  for (auto& a : sc->pools)
    if (a.stream == stream) area = &a;
  if (!area) {
    if (sc->pools.size() >= 8) {  // a caller that keeps creating streams: drop the oldest area (hipFree waits for its launches)
      (void)hipFree(sc->pools.front().p);
      sc->pools.erase(sc->pools.begin());
    }
    sc->pools.push_back({stream, nullptr, 0});
    area = &sc->pools.back();
  }
  if (need > area->bytes) {
    (void)hipFree(area->p);  // waits for launches that may still use it
    area->p = nullptr, area->bytes = 0;
    HIPCHECK(hipMalloc((void**)&area->p, need));
    area->bytes = need;
  }
  S.pool = area->p;
  return RTC_OK;
}
// After a synchronisation.  RTC_ERR_RAY_POOL: the last launch dropped parked rays (its frame is incomplete); the scene's
// pools are twice as large from now on, so the same call can simply be made again.
static RtcStatus check_pool_flag(RtcScene* sc) {
  if (!sc->d.has_refraction) return RTC_OK;
  unsigned int f = 0;
  HIPCHECK(hipMemcpy(&f, sc->d_pool_flag, sizeof f, hipMemcpyDeviceToHost));
  if (f) {
    (void)hipMemset(sc->d_pool_flag, 0, sizeof f);
    if (sc->pool_scale >= POOL_SCALE_MAX)
      return fail(RTC_ERR_HIP, "internal: a wave's ray pool overflowed at %u x its bound; the frame is incomplete", sc->pool_scale);
    sc->pool_scale *= 2;
    return fail(RTC_ERR_RAY_POOL, "a wave's ray pool overflowed: the frame is incomplete; the pools are now %u x the depth-first bound, render again",
                sc->pool_scale);
  }
  return RTC_OK;
}

// Scheduling hint, never a statement about the picture: the pixel rectangle (global pixel coordinates, half-open) on which a
// pixel can see a top-level group, from the pinhole projection of the groups' boxes.  false: no clean projection.
static bool world_to_camera(const RtcCameraDesc* cam, double a[4][8]) {  // [transform_inverse | I] -> [I | world-to-camera]
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) a[i][j] = cam->transform_inverse[4 * i + j], a[i][4 + j] = i == j ? 1.0 : 0.0;
  for (int c = 0; c < 4; ++c) {
    int piv = c;
    for (int r = c + 1; r < 4; ++r)
      if (std::fabs(a[r][c]) > std::fabs(a[piv][c])) piv = r;
    if (!(std::fabs(a[piv][c]) > 1e-12)) return false;
    for (int j = 0; j < 8; ++j) std::swap(a[c][j], a[piv][j]);
    const double inv = 1.0 / a[c][c];
    for (int j = 0; j < 8; ++j) a[c][j] *= inv;
    for (int r = 0; r < 4; ++r)
      if (r != c) {
        const double f = a[r][c];
        for (int j = 0; j < 8; ++j) a[r][j] -= f * a[c][j];
      }
  }
  return true;
}
static bool project_pixel(const double a[4][8], const RtcCameraDesc* cam, const double w[3], double& px, double& py) {
  double c3[3];
  for (int i = 0; i < 3; ++i) c3[i] = a[i][4] * w[0] + a[i][5] * w[1] + a[i][6] * w[2] + a[i][7];
  if (!(c3[2] < -1e-9)) return false;  // at or behind the camera plane (the canvas is at z = -1)
  // (one division instead of four: the callers widen what they get by a pixel and more, 1e-16 of rounding does not matter)
  const double iz = 1.0 / -c3[2], ips = 1.0 / cam->pixel_size;
  px = (cam->half_width - c3[0] * iz) * ips, py = (cam->half_height - c3[1] * iz) * ips;
  return px == px && py == py;
}
static bool busy_pixel_rect(const RtcScene* sc, const RtcCameraDesc* cam, const double a[4][8], double& x0, double& y0, double& x1, double& y1) {
  if (sc->top_boxes.empty()) return false;
  x0 = y0 = 1e300, x1 = y1 = -1e300;
  for (const auto& b : sc->top_boxes)
    for (int k = 0; k < 8; ++k) {
      const double w[3] = {(k & 1) ? b[3] : b[0], (k & 2) ? b[4] : b[1], (k & 4) ? b[5] : b[2]};
      double px, py;
      if (!project_pixel(a, cam, w, px, py)) return false;
      x0 = std::min(x0, px), x1 = std::max(x1, px), y0 = std::min(y0, py), y1 = std::max(y1, py);
    }
  return true;
}

// RTC_STRIP_ORDER.  Scheduling hint, never a statement about the picture: which workgroup of a launch renders which tile.
// A launch's workgroups are dispatched in order, every eighth one to the same XCD, and whichever long-lived waves start last are
// the frame's tail (tools/wave_timeline.py: with the tiles taken row by row the longest wave of a C3 frame lives 95 of its 102 us,
// and waves of 35 us that start at 65 us end it).  So the host estimates every tile's wave life -- a wave lives longer the more
// clusters (<= 8 triangles) its 64 pixels see: life ~ 7 us + 2.8 us per cluster whose projected box meets the tile (measured on
// C3 and C4: correlation 0.89 / 0.84), times what a hit can spawn (reflective: one ray per generation; transparent: a binary
// tree) -- and builds a table with one entry per strip of 16 tiles of a tile row (neighbouring tiles meet the same clusters and
// share an XCD's L2), sorted by descending life of the strip's longest wave and dealt to the eight XCDs so that the estimated
// loads are equal.  (Measured and dropped, profiles/r03/v3_*: raising the issue priority of the longest waves with s_setprio --
// no change, they wait on their own dependent instructions, not for issue slots -- and rendering the longest tiles as four
// waves of 4 x 4 pixels: such a wave still lives 0.83 of the whole tile's, its triangles all lie within a few pixels.)
// Tiles no cluster projects onto (most of them end at the background test) come last, in row-major order, and run while the
// last long waves drain.  The table is a pure function of (camera, rows of the call, band, recursion limit) and is kept per
// stream, like the ray pools: launches on one stream run in order, so a table may be rewritten by a copy on its own stream.
constexpr size_t STRIP_TABLES_PER_STREAM = 4, STRIP_TABLES_MAX = 16;
static float env_float(const char* name, float dflt) {  // thresholds of the tile order (experiments; the defaults are what was measured)
  const char* e = std::getenv(name);
  return e && *e ? (float)std::atof(e) : dflt;
}
static RtcStatus strip_table(RtcScene* sc, const RtcCameraDesc* cam, const RowMap& R, uint32_t row0, uint32_t tiles_x, uint32_t tiles_y,
                             uint32_t side, int L, hipStream_t stream, const uint32_t** tab, uint32_t* n_out) {
  static_assert(RTC_XCD_STRIP == 16, "table entries describe 16 workgroups");
  *tab = nullptr, *n_out = 0;
  if (sc->cluster_costs.empty() || tiles_x > 0xfffu || tiles_y > 0x1fffu) return RTC_OK;  // (an item has 12 + 13 bits for a tile)
  const uint32_t n_sx = (tiles_x + RTC_XCD_STRIP - 1) / RTC_XCD_STRIP;
  if ((uint64_t)n_sx * tiles_y > (1u << 22)) return RTC_OK;
  const uint32_t n_strips = n_sx * tiles_y;
  std::vector<unsigned char> key;
  auto put = [&](const void* p, size_t n) { key.insert(key.end(), (const unsigned char*)p, (const unsigned char*)p + n); };
  put(&cam->hsize, sizeof cam->hsize), put(&cam->vsize, sizeof cam->vsize), put(&cam->half_width, sizeof(double));
  put(&cam->half_height, sizeof(double)), put(&cam->pixel_size, sizeof(double)), put(cam->transform_inverse, 16 * sizeof(double));
  const uint32_t geo[8] = {R.block_rows, R.block_first, R.block_step, R.local_rows, row0, tiles_x, tiles_y, (uint32_t)L};
  put(geo, sizeof geo);
  RtcScene::StripTable* hit = nullptr;
  size_t on_stream = 0;
  for (auto& t : sc->strip_tables)
    if (t.uploaded_on == stream) {
      ++on_stream;
      if (t.key == key) hit = &t;
    }
  if (hit) {
    hit->used = ++sc->strip_clock;
    sc->tables_unused_in_a_row = 0;
    *tab = hit->dev, *n_out = (uint32_t)hit->n;
    return RTC_OK;
  }
  {  // a table is a hint: a launch without one takes its tiles in the order of round 2 and traces every tile
    uint64_t h = 1469598103934665603ull;
    for (unsigned char c : key) h = (h ^ c) * 1099511628211ull;
    h |= 1ull;  // (zero marks an empty place of the ring)
    bool seen = false;
    for (uint64_t r : sc->views_without_table) seen = seen || r == h;
    const char* always = std::getenv("RTC_STRIP_ALWAYS");  // experiments: "1" builds a table for every new view
    if (!seen && sc->tables_unused_in_a_row >= 3 && !(always && always[0] == '1')) {
      sc->views_without_table[sc->views_without_table_at++ % sc->views_without_table.size()] = h;
      return RTC_OK;
    }
    ++sc->tables_unused_in_a_row;
  }
  // ---- estimate: wave life per tile, in microseconds of a full machine --------------------------------------------------
  static const bool phase_times = std::getenv("RTC_STRIP_TIMING") != nullptr;  // diagnostics: host time of the phases below, on stderr
  auto now_us = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double ph0 = phase_times ? now_us() : 0.0;
  double ph1 = 0.0, ph2 = 0.0, ph3 = 0.0, ph4 = 0.0;
  double w2c[4][8];
  if (!world_to_camera(cam, w2c)) return RTC_OK;
  const bool whole_frame = (R.block_rows == 0 || R.block_rows >= cam->vsize) && R.block_first == 0 && R.block_step == 1;
  auto local_row = [&](double gy) -> double {  // global pixel row -> row of this call's dense output
    if (!(gy > 0.0)) return 0.0;
    const uint64_t g = (uint64_t)std::min(gy, (double)cam->vsize);
    if (whole_frame) return (double)std::min<uint64_t>(g, R.local_rows);  // (the general case below costs four integer divisions)
    const uint64_t B = R.block_rows ? R.block_rows : cam->vsize, nb = g / B, rem = g % B;
    const uint64_t k = nb <= R.block_first ? 0 : (nb - R.block_first + R.block_step - 1) / R.block_step;  // owned blocks before nb
    const bool owned = nb >= R.block_first && (nb - R.block_first) % R.block_step == 0;
    return (double)std::min<uint64_t>(k * B + (owned ? rem : 0), R.local_rows);
  };
  const double s_ = (double)side;
  // pixel box -> tile rectangle of this band (inclusive); false: none of its rows belongs to this call / band
  auto tile_rect = [&](double bx0, double by0, double bx1, double by1, long long& tx0, long long& ty0, long long& tx1, long long& ty1) {
    if (bx1 < 0.0 || by1 < 0.0 || bx0 >= (double)cam->hsize || by0 >= (double)cam->vsize) return false;
    const double ly0 = local_row(by0), ly1 = local_row(by1 + 1.0);
    if (!(ly1 > ly0)) return false;
    tx0 = std::max(0ll, (long long)std::floor(bx0 / s_)), tx1 = std::min((long long)tiles_x - 1, (long long)std::floor(bx1 / s_));
    ty0 = std::max(0ll, (long long)std::floor(ly0 / s_) - (long long)row0);
    ty1 = std::min((long long)tiles_y - 1, (long long)std::floor((ly1 - 1.0) / s_) - (long long)row0);
    return tx1 >= tx0 && ty1 >= ty0;
  };
  const int G = std::max(1, (L + 2) / 3);
  const float factor[3] = {1.f, (float)G, (float)((1u << std::min(G, 20)) - 1u)};
  constexpr float LIFE_BASE = 7.f, LIFE_PER_CLUSTER = 2.8f, LIFE_MISS = 5.f, LIFE_BACKGROUND = 2.f;
  std::vector<float> est((size_t)tiles_x * tiles_y, 0.f);
  // RTC_CLEAR_EMPTY: covered[t] = some leaf's bounds may be met by a ray of tile t.  This one IS a statement about the picture: a
  // tile that is not covered is stored BLACK without being traced (a ray that meets no leaf's bounds has no intersection, and
  // World::color_at of it is BLACK: world.rs:93), so the projection is conservative: the pixel box of a bound's eight projected
  // corners (a convex body projects into the hull of its corners' projections) widened by a pixel; a corner at or behind the
  // camera's plane, a leaf without bounds (plane, untruncated cylinder / cone) or a camera that cannot be inverted: no tile is
  // declared empty.  A pixel's ray passes through its centre (camera.rs:52-53), i.e. through pixel coordinate x + 0.5 here.
  std::vector<unsigned char> covered((size_t)tiles_x * tiles_y, 0);
  bool claims = RTC_CLEAR_EMPTY && !sc->unbounded_leaf;
  auto cover = [&](const RtcScene::BlockCost& bc, float v, bool mark, double* box) -> bool {  // false: no pixel box (no claims any more)
    double bx0 = 1e300, bx1 = -1e300, by0 = 1e300, by1 = -1e300;
    bool ok = true;
    for (int k = 0; k < 8 && ok; ++k) {
      double px, py;
      ok = project_pixel(w2c, cam, bc.corner[k], px, py);
      if (ok) bx0 = std::min(bx0, px), bx1 = std::max(bx1, px), by0 = std::min(by0, py), by1 = std::max(by1, py);
    }
    if (!ok) {
      claims = false;
      return false;
    }
    if (box) box[0] = bx0, box[1] = bx1, box[2] = by0, box[3] = by1;
    long long tx0, ty0, tx1, ty1;
    if (v > 0.f && tile_rect(bx0, by0, bx1, by1, tx0, ty0, tx1, ty1))
      for (long long ty = ty0; ty <= ty1; ++ty)
        for (long long tx = tx0; tx <= tx1; ++tx) est[(size_t)ty * tiles_x + (size_t)tx] += v;
    // widened by a pixel, and by half a percent of the box's own extent: check_axis treats a direction component below EPSILON as
    // parallel (shape.rs:593-599), so a ray counted as inside a cube's slab may have drifted out of it by 1e-5 per unit of t
    const double m = 1.0 + 0.005 * std::max(bx1 - bx0, by1 - by0) + 1e-9 * (std::fabs(bx0) + std::fabs(bx1) + std::fabs(by0) + std::fabs(by1));
    if (claims && mark && tile_rect(bx0 - m, by0 - m, bx1 + m, by1 + m, tx0, ty0, tx1, ty1))
      for (long long ty = ty0; ty <= ty1; ++ty)
        for (long long tx = tx0; tx <= tx1; ++tx) covered[(size_t)ty * tiles_x + (size_t)tx] = 1;
    return true;
  };
  // RTC_TRI_COVER: with the triangles' own vertices at hand a cluster box only rates the tiles it projects onto; which of them a
  // ray can meet a triangle on is decided triangle by triangle (the same conservative pixel box, of three projected corners: a
  // triangle in front of the camera's plane projects into the hull of its corners' projections; Shape::intersect's Triangle arm
  // accepts 0 <= u, v, u + v <= 1 of p1 + u e1 + v e2 and nothing else, shape.rs:447-455).
  // A cluster whose (widened) pixel box lies on tiles that are all covered already has nothing to add -- its triangles' boxes lie inside
  // its own -- and is skipped: most of a mesh's interior.
  const bool by_triangle = claims && !sc->tri_world.empty();
  for (const auto& bc : sc->cluster_costs) {
    double cb[4];
    if (!cover(bc, LIFE_PER_CLUSTER * factor[bc.mat_class], !by_triangle, cb) || !by_triangle || !claims) continue;
    {  // (one more pixel and four times the relative term: every triangle's widened box lies inside this one)
      const double m = 2.0 + 0.005 * std::max(cb[1] - cb[0], cb[3] - cb[2]) + 5e-9 * (std::fabs(cb[0]) + std::fabs(cb[1]) + std::fabs(cb[2]) + std::fabs(cb[3]));
      long long tx0, ty0, tx1, ty1;
      if (!tile_rect(cb[0] - m, cb[2] - m, cb[1] + m, cb[3] + m, tx0, ty0, tx1, ty1)) continue;  // (none of this call's tiles)
      bool all = true;
      for (long long ty = ty0; ty <= ty1 && all; ++ty)
        for (long long tx = tx0; tx <= tx1 && all; ++tx) all = covered[(size_t)ty * tiles_x + (size_t)tx] != 0;
      if (all) continue;
    }
    for (size_t t = (size_t)bc.tri_first * 9; t < ((size_t)bc.tri_first + bc.tri_count) * 9; t += 9) {
      double bx0 = 1e300, bx1 = -1e300, by0 = 1e300, by1 = -1e300;
      bool ok = true;
      for (int k = 0; k < 3 && ok; ++k) {
        double px, py;
        ok = project_pixel(w2c, cam, &sc->tri_world[t + 3 * (size_t)k], px, py);
        if (ok) bx0 = std::min(bx0, px), bx1 = std::max(bx1, px), by0 = std::min(by0, py), by1 = std::max(by1, py);
      }
      if (!ok) {  // a corner at or behind the camera's plane: no statement about any tile
        claims = false;
        break;
      }
#ifdef RTC_TRI_COVER_BREAK  // the deliberately wrong variant the tests must catch: the pixel box shrunk instead of widened
      const double m = -1.5;
#else
      const double m = 1.0 + 0.005 * std::max(bx1 - bx0, by1 - by0) + 1e-9 * (std::fabs(bx0) + std::fabs(bx1) + std::fabs(by0) + std::fabs(by1));
#endif
      long long tx0, ty0, tx1, ty1;
      if (tile_rect(bx0 - m, by0 - m, bx1 + m, by1 + m, tx0, ty0, tx1, ty1))
        for (long long ty = ty0; ty <= ty1; ++ty)
          for (long long tx = tx0; tx <= tx1; ++tx) covered[(size_t)ty * tiles_x + (size_t)tx] = 1;
    }
  }
  if (phase_times) ph1 = now_us();
  for (const auto& bc : sc->prim_bounds) (void)cover(bc, 0.f, true, nullptr);
  if (phase_times) ph2 = now_us();
  {  // tiles without a cluster: inside the projection of the top-level group boxes a wave sweeps and misses, outside it leaves at once
    double px0, py0, px1, py1;
    long long tx0 = 0, ty0 = 0, tx1 = -1, ty1 = -1;
    if (busy_pixel_rect(sc, cam, w2c, px0, py0, px1, py1)) {
      if (!tile_rect(px0, py0, px1, py1, tx0, ty0, tx1, ty1)) tx1 = -1, ty1 = -1;
    } else {
      tx0 = ty0 = 0, tx1 = (long long)tiles_x - 1, ty1 = (long long)tiles_y - 1;  // no clean projection: no statement
    }
    for (uint32_t ty = 0; ty < tiles_y; ++ty)
      for (uint32_t tx = 0; tx < tiles_x; ++tx) {
        float& e = est[(size_t)ty * tiles_x + tx];
        if (e > 0.f) e += LIFE_BASE;
        else e = ((long long)tx >= tx0 && (long long)tx <= tx1 && (long long)ty >= ty0 && (long long)ty <= ty1) ? LIFE_MISS : LIFE_BACKGROUND;
      }
  }
#if RTC_WAVE_TIMELINE
  if (const char* f = std::getenv("RTC_STRIP_COST_DUMP")) {  // diagnostics: the per-tile estimate (tiles_x * tiles_y floats)
    if (FILE* fp = std::fopen(f, "wb")) {
      std::fwrite(est.data(), sizeof(float), est.size(), fp);
      std::fwrite(covered.data(), 1, covered.size(), fp);
      const unsigned char c = claims ? 1 : 0;
      std::fwrite(&c, 1, 1, fp);
      std::fclose(fp);
    }
  }
#endif
  // ---- entries ----------------------------------------------------------------------------------------------------------
  // An entry = the work items of one strip of sixteen tiles (neighbouring tiles meet the same clusters and share an XCD's L2); an
  // item = what one workgroup does (RowMap::strip_tab).  Tiles whose estimate exceeds RTC_SPLIT_ABOVE / RTC_SPLIT4_ABOVE of the
  // frame's (sum of the estimates / wave slots of the device) become two items of 8 x 4 / four items of 4 x 4 pixels: the frame cannot
  // end before its longest wave does, and a wave of half (a quarter of) the tile lives about 0.85 (0.75) of the whole tile's -- it
  // meets nearly the same triangles: the gain is small, but it is the bound.  Only in scenes without a transparent material, where a
  // pixel's value does not depend on which wave renders it.
  struct Entry {
    uint32_t first = 0, count = 0;  // its items in `flat` (one allocation for all entries: a new view is built per frame by a moving camera)
    float key = 0.f, load = 0.f;    // life of its longest wave, sum of its waves' lives
  };
  std::vector<Entry> entries;
  entries.reserve(2 * (size_t)n_strips + est.size() / 256 + 16);
  std::vector<uint32_t> flat;
  flat.reserve(est.size() + est.size() / 8 + 64);
  std::vector<uint32_t> clear_list;  // tiles (item format) that are stored BLACK without being traced, row-major
  clear_list.reserve(est.size());
  double total = 0.0;
  for (size_t t = 0; t < est.size(); ++t)
    if (!(claims && !covered[t])) total += est[t];
  static const float split2_at = env_float("RTC_SPLIT_ABOVE", (float)RTC_SPLIT_ABOVE), split4_at = env_float("RTC_SPLIT4_ABOVE", (float)RTC_SPLIT4_ABOVE);
  const float frame_est = (float)(total / 3072.0);
  const float split_above = sc->d.has_refraction ? INFINITY : std::max(30.f, split2_at * frame_est);
  const float split4_above = sc->d.has_refraction ? INFINITY : std::max(45.f, split4_at * frame_est);
  constexpr float HALF_LIFE = 0.9f, QUARTER_LIFE = 0.8f;
  auto push_entry = [&](const uint32_t* items, uint32_t n, float key, float load_) {
    Entry x;
    x.first = (uint32_t)flat.size(), x.count = n, x.key = key, x.load = load_;
    flat.insert(flat.end(), items, items + n);
    entries.push_back(x);
  };
  for (uint32_t ty = 0; ty < tiles_y; ++ty)
    for (uint32_t sx = 0; sx < n_sx; ++sx) {
      const uint32_t x0 = sx * RTC_XCD_STRIP;
      // whole tiles of the strip; the parts of its split tiles (an entry of their own: dispatched earlier)
      uint32_t e_items[RTC_XCD_STRIP], p_items[4 * RTC_XCD_STRIP], ne = 0, np = 0;
      float e_key = 0.f, e_load = 0.f, p_key = 0.f, p_load = 0.f;
      for (uint32_t w = 0; w < RTC_XCD_STRIP && x0 + w < tiles_x; ++w) {
        const float t = est[(size_t)ty * tiles_x + x0 + w];
        const uint32_t tile = (x0 + w) | ty << 12;
        if (claims && !covered[(size_t)ty * tiles_x + x0 + w]) {  // no leaf can be met there: BLACK, stored by a clearing wave
          clear_list.push_back(tile);
        } else if (t > split4_above) {
          for (uint32_t q = 0; q < 4; ++q) p_items[np++] = tile | (3u + q) << 25;
          p_key = std::max(p_key, QUARTER_LIFE * t), p_load += 4.f * QUARTER_LIFE * t;
        } else if (t > split_above) {
          for (uint32_t q = 0; q < 2; ++q) p_items[np++] = tile | (1u + q) << 25;
          p_key = std::max(p_key, HALF_LIFE * t), p_load += 2.f * HALF_LIFE * t;
        } else {
          e_items[ne++] = tile, e_key = std::max(e_key, t), e_load += t;
        }
      }
      if (np) push_entry(p_items, np, p_key, p_load);
      if (ne) push_entry(e_items, ne, e_key, e_load);
    }
  std::stable_sort(entries.begin(), entries.end(), [](const Entry& a, const Entry& b) { return a.key > b.key; });
  // clearing items (part 7): one per sixteen tiles of the list behind the table.  Last: they cost next to nothing.
  for (size_t i = 0; i < clear_list.size(); i += 256) {
    uint32_t c_items[16], nc = 0;
    for (size_t g = i / 16; g < (std::min(clear_list.size(), i + 256) + 15) / 16; ++g) c_items[nc++] = (uint32_t)g | 7u << 25;
    push_entry(c_items, nc, 0.f, 0.2f * (float)nc);
  }
  // dealt to the XCDs: workgroup w of a launch goes to XCD w % 8, so item k of XCD x's list is table position 8 k + x; the next entry
  // goes to the XCD with the least estimated load so far (ties: the lowest number, i.e. round-robin among equals).  An XCD's
  // entries follow one another without gaps: no workgroup of the launch returns at once but the padding at the very end (workgroups
  // that return at once at the head of a launch keep the dispatcher from filling the machine: 82 instead of 99 % of the wave
  // slots taken during the first 30 us of a C3 frame, profiles/r03/v3_* item 10).
  std::vector<uint32_t> lists[8];
  for (auto& l : lists) l.reserve(flat.size() / 6 + 64);
  double load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (const Entry& e : entries) {
    int x = 0;
    for (int k = 1; k < 8; ++k)
      if (load[k] < load[x]) x = k;
    lists[x].insert(lists[x].end(), flat.begin() + e.first, flat.begin() + e.first + e.count), load[x] += e.load;
  }
  if (phase_times) ph3 = now_us();
  size_t depth = 0;
  for (int x = 0; x < 8; ++x) depth = std::max(depth, lists[x].size());
  const uint32_t n_entries = (uint32_t)(depth * 8);  // items, padding included
  // ---- a slot for it ----------------------------------------------------------------------------------------------------
  RtcScene::StripTable* slot = nullptr;
  if (on_stream >= STRIP_TABLES_PER_STREAM) {  // rewrite this stream's least recently used table (in order behind its last launch)
    for (auto& t : sc->strip_tables)
      if (t.uploaded_on == stream && (!slot || t.used < slot->used)) slot = &t;
  } else {
    if (sc->strip_tables.size() >= STRIP_TABLES_MAX) {  // a caller that keeps creating streams: drop the oldest (hipFree waits for its launches)
      size_t o = 0;
      for (size_t i = 1; i < sc->strip_tables.size(); ++i)
        if (sc->strip_tables[i].used < sc->strip_tables[o].used) o = i;
      (void)hipFree(sc->strip_tables[o].dev), (void)hipHostFree(sc->strip_tables[o].host);
      if (sc->strip_tables[o].copied) (void)hipEventDestroy(sc->strip_tables[o].copied);
      sc->strip_tables.erase(sc->strip_tables.begin() + (long)o);
    }
    sc->strip_tables.emplace_back();
    slot = &sc->strip_tables.back();
    slot->uploaded_on = stream;
  }
  if (!slot->copied) HIPCHECK(hipEventCreateWithFlags(&slot->copied, hipEventDisableTiming));
  else HIPCHECK(hipEventSynchronize(slot->copied));  // (the copy of the table this slot held: several launches ago)
  slot->key.clear(), slot->n = 0;
  const size_t n_units = (size_t)n_entries + 1 + clear_list.size();  // the items, the number of tiles to clear, their list
  if (slot->cap < n_units) {
    (void)hipFree(slot->dev), (void)hipHostFree(slot->host);  // (hipFree waits for launches that may still read it)
    slot->dev = slot->host = nullptr, slot->cap = 0;
    HIPCHECK(hipMalloc((void**)&slot->dev, n_units * sizeof(uint32_t)));
    HIPCHECK(hipHostMalloc((void**)&slot->host, n_units * sizeof(uint32_t), hipHostMallocDefault));
    slot->cap = n_units;
  }
  for (size_t i = 0; i < n_entries; ++i) slot->host[i] = 0xffffffffu;
  for (int x = 0; x < 8; ++x)
    for (size_t k = 0; k < lists[x].size(); ++k) slot->host[k * 8 + (size_t)x] = lists[x][k];
  slot->host[n_entries] = (uint32_t)clear_list.size();
  if (!clear_list.empty()) std::memcpy(slot->host + n_entries + 1, clear_list.data(), clear_list.size() * sizeof(uint32_t));
  HIPCHECK(hipMemcpyAsync(slot->dev, slot->host, n_units * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
  HIPCHECK(hipEventRecord(slot->copied, stream));
  if (phase_times) {
    ph4 = now_us();
    std::fprintf(stderr, "strip_table: clusters %.0f us, triangles + primitives %.0f us, entries + order %.0f us, slot + copy %.0f us (%u items, %zu cleared tiles)\n",
                 ph1 - ph0, ph2 - ph1, ph3 - ph2, ph4 - ph3, n_entries, clear_list.size());
  }
  slot->key = std::move(key), slot->n = n_entries;
  slot->used = ++sc->strip_clock;
  *tab = slot->dev, *n_out = n_entries;
  return RTC_OK;
}

// One launch of the render kernel for the rows R describes (RowMap: which 16-row blocks, how many dense rows).
// Everything is validated by the callers; asynchronous on `stream`.
static RtcStatus launch_render(RtcScene* sc, const RtcCameraDesc* cam, const RowMap& R, int L, uint32_t flags, float* d_rgb,
                               double* d_rgb64, hipStream_t stream) {
  if (R.local_rows == 0) return RTC_OK;
  DCamera C;
  C.hsize = cam->hsize, C.vsize = cam->vsize;
  C.half_width = cam->half_width, C.half_height = cam->half_height, C.pixel_size = cam->pixel_size;
  std::memcpy(C.inv, cam->transform_inverse, sizeof C.inv);
  for (int i = 0; i < 3; ++i) {  // camera.rs:57-59, the pixel-invariant part (this file is compiled with contraction off)
    const double* row = C.inv + 4 * i;
    C.origin[i] = row[0] * 0.0 + row[1] * 0.0 + row[2] * 0.0 + row[3] * 1.0;
    C.zt[i] = row[2] * -1.0;
    C.wt[i] = row[3] * 1.0;
  }
  DScene S = sc->d;
  S.counters = (flags & RTC_FLAG_COUNT_RAYS) ? sc->d_counters : nullptr;
  const bool accel = sc->accel_ok && !(flags & RTC_FLAG_EXACT_SWEEP);
  const uint32_t side = (uint32_t)tile_side(accel);
  const uint32_t tile_rows = (R.local_rows + side - 1) / side, tiles_x = (cam->hsize + side - 1) / side;
  const bool count = S.counters != nullptr;
  // Kernels with a global ray-pool area get it per wave of the launch: a frame whose area would exceed POOL_BUDGET is
  // launched in bands of tile rows that reuse one area (in order, on the same stream).
  uint32_t band_rows = tile_rows;
  {
    const uint64_t per_wave = (uint64_t)pool_entries(sc, L) * 64;
    const uint64_t waves_per_row = (uint64_t)tiles_x * (uint64_t)(block_threads(accel) / 64);
    const uint64_t budget = pool_budget();
    if (per_wave && per_wave * waves_per_row * tile_rows > budget)
      band_rows = (uint32_t)std::max<uint64_t>(1, budget / (per_wave * waves_per_row));
  }
  RowMap Rb = R;
  for (uint32_t row0 = 0; row0 < tile_rows; row0 += band_rows) {
  Rb.tile_row0 = R.tile_row0 + row0;
  dim3 grid(tiles_x, std::min(band_rows, tile_rows - row0));
  Rb.tiles_x = grid.x, Rb.tiles_y = grid.y;
  Rb.rx0 = Rb.ry0 = Rb.rx1 = Rb.ry1 = Rb.n_first = 0, Rb.n_order = 0;
  Rb.strip_tab = nullptr, Rb.n_strips = 0;
#if RTC_XCD_STRIP
#if RTC_STRIP_ORDER
  if (accel) {
    RtcStatus st = strip_table(sc, cam, R, row0, grid.x, grid.y, side, L, stream, &Rb.strip_tab, &Rb.n_strips);
    if (st != RTC_OK) return st;
  }
#endif
  if (Rb.n_strips) {
    grid = dim3((unsigned)Rb.n_strips, 1);  // (one workgroup per item; a multiple of 8)
  } else {
    const uint64_t unit = 8ull * RTC_XCD_STRIP;
    uint64_t n_first = 0;
#if RTC_BUSY_FIRST
    // Tiles that can see a mesh first, the others (most of which leave at the background test) after them: a launch's workgroups
    // are dispatched in order, and a wave that only finds out that its tile is background still holds a slot while it does;
    // queued behind the busy tiles those waves run while the long-lived ones drain instead of between them.
    double px0, py0, px1, py1, w2c[4][8];
    Rb.n_order = 0;
    if (accel && world_to_camera(cam, w2c) && busy_pixel_rect(sc, cam, w2c, px0, py0, px1, py1)) {
      // global pixel rows -> rows of this call's dense output (the call may own every block_step-th block of block_rows rows)
      auto local_row = [&](double gy) -> double {
        if (!(gy > 0.0)) return 0.0;
        const uint64_t g = (uint64_t)std::min(gy, (double)cam->vsize);
        const uint64_t B = R.block_rows ? R.block_rows : cam->vsize, nb = g / B, rem = g % B;
        const uint64_t k = nb <= R.block_first ? 0 : (nb - R.block_first + R.block_step - 1) / R.block_step;  // owned blocks before nb
        const bool owned = nb >= R.block_first && (nb - R.block_first) % R.block_step == 0;
        return (double)std::min<uint64_t>(k * B + (owned ? rem : 0), R.local_rows);
      };
      const double s_ = (double)side, ly0 = local_row(py0), ly1 = local_row(py1 + 1.0);
      const long long tx0 = (long long)std::floor(px0 / s_) - 1, ty0 = (long long)std::floor(ly0 / s_) - 1 - (long long)row0;
      const long long tx1 = (long long)std::floor(px1 / s_) + 2, ty1 = (long long)std::floor(ly1 / s_) + 2 - (long long)row0;
      const long long cx0 = std::max(0ll, tx0), cy0 = std::max(0ll, ty0), cx1 = std::min((long long)grid.x, tx1), cy1 = std::min((long long)grid.y, ty1);
      if (cx1 > cx0 && cy1 > cy0 && (uint64_t)(cx1 - cx0) * (cy1 - cy0) < (uint64_t)grid.x * grid.y) {
        Rb.rx0 = (uint32_t)cx0, Rb.ry0 = (uint32_t)cy0, Rb.rx1 = (uint32_t)cx1, Rb.ry1 = (uint32_t)cy1;
        n_first = ((uint64_t)(cx1 - cx0) * (cy1 - cy0) + unit - 1) / unit * unit;
        Rb.n_first = (uint32_t)n_first;
#if RTC_ROW_COST_ORDER
        // Longest waves first: a wave's life grows with the number of triangles its 64 pixels see, so the rectangle's tile rows are
        // taken by descending triangle density of the mesh tiles (<= 512 triangles each) that project onto them; ties top-down.
        const uint32_t rh = (uint32_t)(cy1 - cy0);
        if (rh <= 272 && !sc->cost_boxes.empty()) {
          std::vector<double> wrow(rh, 0.0);
          for (const auto& cb : sc->cost_boxes) {
            double bx0 = 1e300, bx1 = -1e300, by0 = 1e300, by1 = -1e300;
            bool ok = true;
            for (int k = 0; k < 8 && ok; ++k) {
              double px, py;
              ok = project_pixel(w2c, cam, cb.corner[k], px, py);
              if (ok) bx0 = std::min(bx0, px), bx1 = std::max(bx1, px), by0 = std::min(by0, py), by1 = std::max(by1, py);
            }
            if (!ok) continue;
            const double area = std::max(1.0, bx1 - bx0) * std::max(1.0, by1 - by0), density = cb.tris / area;
            const long long r0 = (long long)std::floor(local_row(by0) / s_) - (long long)row0 - cy0;
            const long long r1 = (long long)std::floor(local_row(by1 + 1.0) / s_) - (long long)row0 - cy0;
            for (long long r = std::max(0ll, r0); r <= std::min((long long)rh - 1, r1); ++r) wrow[(size_t)r] = std::max(wrow[(size_t)r], density);
          }
          std::vector<uint16_t> ord(rh);
          for (uint32_t r = 0; r < rh; ++r) ord[r] = (uint16_t)r;
          std::stable_sort(ord.begin(), ord.end(), [&](uint16_t x, uint16_t y) { return wrow[x] > wrow[y]; });
          for (uint32_t r = 0; r < rh; ++r) Rb.row_order[r] = ord[r];
          Rb.n_order = rh;
        }
#endif
      }
    }
#endif
    const uint64_t rest = (uint64_t)grid.x * grid.y - (uint64_t)(Rb.rx1 - Rb.rx0) * (Rb.ry1 - Rb.ry0);
    grid = dim3((unsigned)(n_first + (rest + unit - 1) / unit * unit), 1);
  }
#endif
  {
    RtcStatus st = ensure_pool(sc, L, (uint64_t)grid.x * grid.y * (uint64_t)(block_threads(accel) / 64), !accel || S.n_tri != 0, S, stream);
    if (st != RTC_OK) return st;
  }
#define RTC_LAUNCH_RENDER(REFR, ACCEL, COUNT, TRIS, POOLG) \
  hipLaunchKernelGGL((render_kernel<REFR, ACCEL, COUNT, TRIS, POOLG>), grid, dim3(block_threads(ACCEL)), 0, stream, S, C, Rb, L, d_rgb, d_rgb64)
#define RTC_LAUNCH_RENDER_C(REFR, ACCEL, TRIS, POOLG) \
  do { if (count) RTC_LAUNCH_RENDER(REFR, ACCEL, true, TRIS, POOLG); else RTC_LAUNCH_RENDER(REFR, ACCEL, false, TRIS, POOLG); } while (0)
  // the accelerated kernels exist with and without the triangle machinery (a scene of primitives only runs in
  // ~100 instead of 168 VGPRs); the brute-force kernels are not specialised
  const bool tris = S.n_tri != 0;
  if (S.has_refraction) {
    // three generations or more (L >= 7): parked rays can outnumber the 64 a wave keeps in LDS, so those kernels carry
    // the overflow path to global memory (whose stores cost the loop its scalar loads: +3 % on C5, +15 % on C2, hence
    // a variant of its own); up to two generations the LDS pool always suffices
    if (S.pool_cap) {
      if (!accel) RTC_LAUNCH_RENDER_C(true, false, true, true);
      else if (tris) RTC_LAUNCH_RENDER_C(true, true, true, true);
      else RTC_LAUNCH_RENDER_C(true, true, false, true);
    } else {
      if (!accel) RTC_LAUNCH_RENDER_C(true, false, true, false);
      else if (tris) RTC_LAUNCH_RENDER_C(true, true, true, false);
      else RTC_LAUNCH_RENDER_C(true, true, false, false);
    }
  } else {
    if (!accel) RTC_LAUNCH_RENDER_C(false, false, true, false);
    else if (tris) RTC_LAUNCH_RENDER_C(false, true, true, false);
    else RTC_LAUNCH_RENDER_C(false, true, false, false);
  }
#undef RTC_LAUNCH_RENDER_C
#undef RTC_LAUNCH_RENDER
  HIPCHECK(hipGetLastError());
  }  // bands
  return RTC_OK;
}

static RtcStatus validate_camera(const RtcScene* sc, const RtcCameraDesc* cam, const RtcRenderOpts* opts) {
  if (cam->hsize == 0 || cam->vsize == 0) return fail(RTC_ERR_INVALID_ARG, "empty image %ux%u", cam->hsize, cam->vsize);
  RtcStatus st = validate_limit(sc, opts->recursion_limit);
  if (st != RTC_OK) return st;
  const double* m = cam->transform_inverse;
  if (!(m[12] == 0.0 && m[13] == 0.0 && m[14] == 0.0 && m[15] == 1.0))
    return fail(RTC_ERR_NON_AFFINE, "camera transform_inverse is not affine");
  return RTC_OK;
}

RtcStatus rtc_render_device(RtcScene* sc, const RtcCameraDesc* cam, const RtcRenderOpts* opts, float* d_rgb,
                            double* d_rgb64, void* hip_stream) {
  if (!sc || !cam || !opts || !d_rgb) return fail(RTC_ERR_INVALID_ARG, "null argument");
  RtcStatus st = validate_camera(sc, cam, opts);
  if (st != RTC_OK) return st;
  RowMap R;
  R.block_rows = opts->block_rows ? opts->block_rows : cam->vsize;
  R.block_first = opts->block_rows ? opts->block_first : 0;
  R.block_step = (opts->block_rows && opts->block_step) ? opts->block_step : 1;
  R.local_rows = rtc_local_rows(cam->vsize, opts);
  R.tile_row0 = 0, R.tiles_x = R.tiles_y = 0, R.rx0 = R.ry0 = R.rx1 = R.ry1 = R.n_first = 0, R.n_order = 0;
  if (R.local_rows == 0) return RTC_OK;  // nothing to do for this shard
  HIPCHECK(hipSetDevice(sc->device));
  hipStream_t stream = (hipStream_t)hip_stream;
  sc->timed = (opts->flags & RTC_FLAG_TIMING) != 0;
  sc->last_stream = stream;
  if (sc->timed) HIPCHECK(hipEventRecord(sc->ev0, stream));
  st = launch_render(sc, cam, R, (int)opts->recursion_limit, opts->flags, d_rgb, d_rgb64, stream);
  if (st != RTC_OK) return st;
  if (sc->timed) HIPCHECK(hipEventRecord(sc->ev1, stream));
  return RTC_OK;
}

RtcStatus rtc_scene_stats(RtcScene* sc, RtcStats* stats) {
  if (!sc || !stats) return fail(RTC_ERR_INVALID_ARG, "null argument");
  HIPCHECK(hipSetDevice(sc->device));
  HIPCHECK(hipStreamSynchronize(sc->last_stream));
  unsigned long long c[RTC_N_COUNTERS] = {0};
  HIPCHECK(hipMemcpy(c, sc->d_counters, sizeof c, hipMemcpyDeviceToHost));
  stats->radiance_rays = c[0];
  stats->shadow_rays = c[1];
  stats->triangle_tests = c[2];
  stats->primitive_tests = c[3];
  stats->filter_tests = c[4];
  stats->box_tests = c[5];
  stats->culled_rays = c[20];
  stats->unswept_shadow_rays = c[26];
  stats->kernel_ms = 0.0;
  if (sc->timed) {
    float ms = 0.f;
    HIPCHECK(hipEventElapsedTime(&ms, sc->ev0, sc->ev1));
    stats->kernel_ms = ms;
  }
  stats->upload_ms = sc->upload_ms;
  stats->download_ms = 0.0;
  return check_pool_flag(sc);
}

// Diagnostics (not part of include/rtc_hip.h): the raw counter block, including the region timers of a
// -DRTC_REGION_TIMERS=1 build (all zero otherwise).  out must hold 32 values.
extern "C" RtcStatus rtc_debug_counters(RtcScene* sc, unsigned long long* out) {  // out: RTC_N_COUNTERS (32) values
  if (!sc || !out) return fail(RTC_ERR_INVALID_ARG, "null argument");
  HIPCHECK(hipSetDevice(sc->device));
  HIPCHECK(hipStreamSynchronize(sc->last_stream));
  HIPCHECK(hipMemcpy(out, sc->d_counters, RTC_N_COUNTERS * sizeof(unsigned long long), hipMemcpyDeviceToHost));
#if RTC_REGION_TIMERS == 2
  {
    std::vector<unsigned long long> rows((size_t)TIMER_ROWS * 32);
    HIPCHECK(hipMemcpyFromSymbol(rows.data(), HIP_SYMBOL(g_timers), rows.size() * sizeof(unsigned long long)));
    for (int k = 6; k < 32; ++k) {
      if (k == 20 || k == 26) continue;  // ray counters, not timers
      unsigned long long sum = 0;
      for (int r = 0; r < TIMER_ROWS; ++r) sum += rows[(size_t)r * 32 + k];
      out[k] = sum;
    }
  }
#endif
  return RTC_OK;
}

#if RTC_WAVE_TIMELINE
// Diagnostics: the first n workgroups' records of the render launches since the last call (16 values each; a record whose
// fourth value is zero was not written; TIMELINE_REC = 16 values per record).  Clears them.
extern "C" RtcStatus rtc_debug_timeline(unsigned long long* out, uint32_t n) {
  if (!out || n > TIMELINE_WGS) return fail(RTC_ERR_INVALID_ARG, "bad argument");
  HIPCHECK(hipDeviceSynchronize());
  HIPCHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_timeline), (size_t)n * TIMELINE_REC * sizeof(unsigned long long)));
  void* sym = nullptr;
  HIPCHECK(hipGetSymbolAddress(&sym, HIP_SYMBOL(g_timeline)));
  HIPCHECK(hipMemset(sym, 0, sizeof(unsigned long long) * (size_t)TIMELINE_WGS * TIMELINE_REC));
  return RTC_OK;
}
#endif

RtcStatus rtc_scene_reset_counters(RtcScene* sc) {
  if (!sc) return fail(RTC_ERR_INVALID_ARG, "null argument");
  HIPCHECK(hipSetDevice(sc->device));
  HIPCHECK(hipStreamSynchronize(sc->last_stream));
  // complete before this returns: the next launch may go to a non-blocking stream, which a memset queued on the null
  // stream would not be ordered with
  HIPCHECK(hipMemsetAsync(sc->d_counters, 0, RTC_N_COUNTERS * sizeof(unsigned long long), nullptr));
#if RTC_REGION_TIMERS == 2
  {
    void* sym = nullptr;
    HIPCHECK(hipGetSymbolAddress(&sym, HIP_SYMBOL(g_timers)));
    HIPCHECK(hipMemsetAsync(sym, 0, sizeof(unsigned long long) * TIMER_ROWS * 32, nullptr));
  }
#endif
  HIPCHECK(hipStreamSynchronize(nullptr));
  return RTC_OK;
}

}  // extern "C"

#include "rtc_multi.hpp"  // rtc_render, rtc_multi_*, rtc_render_multi, rtc_release_cached (C linkage from rtc_hip.h)

extern "C" {

RtcStatus rtc_color_at(RtcScene* sc, uint64_t n, const double* origins, const double* directions,
                       const RtcRenderOpts* opts, float* h_rgb, double* h_rgb64) {
  if (!sc || !opts || (n && (!origins || !directions))) return fail(RTC_ERR_INVALID_ARG, "null argument");
  RtcStatus st = validate_limit(sc, opts->recursion_limit);
  if (st != RTC_OK) return st;
  if (n == 0) return RTC_OK;
  HIPCHECK(hipSetDevice(sc->device));
  const size_t bytes = (size_t)n * 3 * sizeof(double);
  if (sc->ray_cap < n) {  // the ray and result buffers stay with the scene and only ever grow
    (void)hipFree(sc->d_ray_o), (void)hipFree(sc->d_ray_d), (void)hipFree(sc->d_ray_out64), (void)hipFree(sc->d_ray_out32);
    sc->d_ray_o = sc->d_ray_d = sc->d_ray_out64 = nullptr, sc->d_ray_out32 = nullptr, sc->ray_cap = 0;
    const size_t cap = std::max<size_t>((size_t)n, 4096);
    if (hipMalloc((void**)&sc->d_ray_o, cap * 24) != hipSuccess || hipMalloc((void**)&sc->d_ray_d, cap * 24) != hipSuccess ||
        hipMalloc((void**)&sc->d_ray_out64, cap * 24) != hipSuccess || hipMalloc((void**)&sc->d_ray_out32, cap * 12) != hipSuccess)
      return fail(RTC_ERR_HIP, "hipMalloc(rays) failed");
    sc->ray_cap = cap;
  }
  double *d_o = sc->d_ray_o, *d_d = sc->d_ray_d, *d_out64 = sc->d_ray_out64;
  float* d_out = sc->d_ray_out32;
  auto cleanup = [&](RtcStatus s) { return s; };
  if (hipMemcpy(d_o, origins, bytes, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(d_d, directions, bytes, hipMemcpyHostToDevice) != hipSuccess)
    return cleanup(fail(RTC_ERR_HIP, "ray upload failed"));
  for (;;) {
  DScene S = sc->d;
  S.counters = (opts->flags & RTC_FLAG_COUNT_RAYS) ? sc->d_counters : nullptr;
  const bool accel = sc->accel_ok && !(opts->flags & RTC_FLAG_EXACT_SWEEP);
  const uint64_t bt = (uint64_t)block_threads(accel);
  dim3 grid((unsigned)((n + bt - 1) / bt));
  st = ensure_pool(sc, (int)opts->recursion_limit, (uint64_t)grid.x * (bt / 64), !accel || S.n_tri != 0, S, nullptr);
  if (st != RTC_OK) return st;
  sc->last_stream = nullptr;
  sc->timed = false;
  const bool count = S.counters != nullptr;
  const int L = (int)opts->recursion_limit;
#define RTC_LAUNCH_COLOR(REFR, ACCEL, COUNT, TRIS, POOLG) \
  hipLaunchKernelGGL((color_at_kernel<REFR, ACCEL, COUNT, TRIS, POOLG>), grid, dim3(block_threads(ACCEL)), 0, nullptr, S, n, d_o, d_d, L, d_out, d_out64)
#define RTC_LAUNCH_COLOR_C(REFR, ACCEL, TRIS, POOLG) \
  do { if (count) RTC_LAUNCH_COLOR(REFR, ACCEL, true, TRIS, POOLG); else RTC_LAUNCH_COLOR(REFR, ACCEL, false, TRIS, POOLG); } while (0)
  const bool tris = S.n_tri != 0;
  if (S.has_refraction) {
    if (S.pool_cap) {
      if (!accel) RTC_LAUNCH_COLOR_C(true, false, true, true);
      else if (tris) RTC_LAUNCH_COLOR_C(true, true, true, true);
      else RTC_LAUNCH_COLOR_C(true, true, false, true);
    } else {
      if (!accel) RTC_LAUNCH_COLOR_C(true, false, true, false);
      else if (tris) RTC_LAUNCH_COLOR_C(true, true, true, false);
      else RTC_LAUNCH_COLOR_C(true, true, false, false);
    }
  } else {
    if (!accel) RTC_LAUNCH_COLOR_C(false, false, true, false);
    else if (tris) RTC_LAUNCH_COLOR_C(false, true, true, false);
    else RTC_LAUNCH_COLOR_C(false, true, false, false);
  }
#undef RTC_LAUNCH_COLOR_C
#undef RTC_LAUNCH_COLOR
  if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) return cleanup(fail(RTC_ERR_HIP, "color_at kernel failed"));
  st = check_pool_flag(sc);
  if (st == RTC_ERR_RAY_POOL) continue;  // a wave's ray pool overflowed: again, with the pools check_pool_flag has doubled
  if (st != RTC_OK) return st;
  break;
  }
  if (h_rgb && hipMemcpy(h_rgb, d_out, (size_t)n * 3 * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess)
    return cleanup(fail(RTC_ERR_HIP, "result copy failed"));
  if (h_rgb64 && hipMemcpy(h_rgb64, d_out64, bytes, hipMemcpyDeviceToHost) != hipSuccess)
    return cleanup(fail(RTC_ERR_HIP, "result copy failed"));
  return cleanup(RTC_OK);
}

RtcStatus rtc_selftest_math(int device, int op, uint64_t n, const double* a, const double* b, double* out) {
  if (!a || !out || op < 0 || op > 6) return fail(RTC_ERR_INVALID_ARG, "bad argument");
  if (device < 0 || device >= rtc_device_count()) return fail(RTC_ERR_NO_DEVICE, "device %d out of range", device);
  if (n == 0) return RTC_OK;
  HIPCHECK(hipSetDevice(device));
  double *d_a = nullptr, *d_b = nullptr, *d_o = nullptr;
  auto cleanup = [&](RtcStatus s) {
    (void)hipFree(d_a), (void)hipFree(d_b), (void)hipFree(d_o);
    return s;
  };
  const size_t bytes = (size_t)n * sizeof(double);
  if (hipMalloc((void**)&d_a, bytes) != hipSuccess || hipMalloc((void**)&d_o, bytes) != hipSuccess ||
      (b && hipMalloc((void**)&d_b, bytes) != hipSuccess))
    return cleanup(fail(RTC_ERR_HIP, "hipMalloc failed"));
  if (hipMemcpy(d_a, a, bytes, hipMemcpyHostToDevice) != hipSuccess ||
      (b && hipMemcpy(d_b, b, bytes, hipMemcpyHostToDevice) != hipSuccess))
    return cleanup(fail(RTC_ERR_HIP, "upload failed"));
  hipLaunchKernelGGL(selftest_math_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, op, n, d_a, d_b, d_o);
  if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) return cleanup(fail(RTC_ERR_HIP, "selftest kernel failed"));
  if (hipMemcpy(out, d_o, bytes, hipMemcpyDeviceToHost) != hipSuccess) return cleanup(fail(RTC_ERR_HIP, "download failed"));
  return cleanup(RTC_OK);
}

}  // extern "C"
