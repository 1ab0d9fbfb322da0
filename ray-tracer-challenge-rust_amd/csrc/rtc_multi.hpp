// rtc_multi.hpp — the host-buffer entry points of include/rtc_hip.h: rtc_render (one device), rtc_multi_* /
// rtc_render_multi (the devices of one node, in one process) and what they keep between calls.
// Included by rtc_hip.hip (it uses RtcScene, launch_render, fail ... of that file); the entry points get their C
// linkage from their declarations in include/rtc_hip.h.
//
// Camera::render (camera.rs:67-80) borrows `&World` for every frame and returns an owned Canvas: a drop-in call
// therefore hands over the whole flattened scene and two host buffers each time.  What makes that cheap here:
//   * scenes stay resident in HBM, keyed by a 128-bit hash of every byte `desc` points to: an identical scene skips
//     the upload and the acceleration build (SceneCache);
//   * the framebuffer comes back through a pinned staging buffer in 4 MiB pieces, the DMA of piece k+1 overlapping
//     the copy of piece k into the caller's (pageable) buffer on a few host threads (Stager);
//   * across devices: interleaved 16-row blocks, each device renders its blocks in chunks and a finished chunk
//     travels to the root device over xGMI (RCCL send/recv, or peer DMA) while the next chunk renders; the root's
//     download then reads the blocks in Canvas order, so no de-interleave pass exists (RtcMulti).
#pragma once

namespace {

// ---- content hash ------------------------------------------------------------------------------------
struct Hash128 {  // two independent multiply-xorshift lanes over 8-byte words; not cryptographic, collision odds ~2^-128
  uint64_t a = 0x9e3779b97f4a7c15ull, b = 0xc2b2ae3d27d4eb4full;
  void word(uint64_t w) {
    a = (a ^ w) * 0xff51afd7ed558ccdull;
    a ^= a >> 32;
    b = (b + w) * 0xc4ceb9fe1a85ec53ull;
    b ^= b >> 29;
  }
  void bytes(const void* p, size_t n) {
    const unsigned char* c = (const unsigned char*)p;
    word(n);
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
      uint64_t w;
      std::memcpy(&w, c + i, 8);
      word(w);
    }
    if (i < n) {
      uint64_t w = 0;
      std::memcpy(&w, c + i, n - i);
      word(w);
    }
  }
  template <class T>
  void array(const T* p, size_t count) {
    if (p && count) bytes(p, count * sizeof(T));
    else word(0x5bd1e995ull);
  }
};

// Second key of a cached scene, compared word for word on a hash match (a collision of the home-made hash above must
// not silently render another resident scene): the counts, the light, every table of at most PROBE_FULL bytes whole, and
// PROBE_WORDS evenly spaced 8-byte words of each larger array.  What is left after both keys agree: two scenes of equal
// shape whose large arrays differ only between the sampled words AND collide in both hash lanes.
struct SceneProbe {
  static constexpr size_t PROBE_FULL = 1024, PROBE_WORDS = 61;
  std::vector<uint64_t> w;
  void bytes(const void* p, size_t n) {
    const unsigned char* c = (const unsigned char*)p;
    w.push_back((uint64_t)n);
    auto word_at = [&](size_t i) {
      uint64_t v = 0;
      std::memcpy(&v, c + i, std::min<size_t>(8, n - i));
      return v;
    };
    if (n <= PROBE_FULL) {
      for (size_t i = 0; i < n; i += 8) w.push_back(word_at(i));
    } else {
      const size_t words = n / 8;
      for (size_t k = 0; k < PROBE_WORDS; ++k) w.push_back(word_at((words - 1) * k / (PROBE_WORDS - 1) * 8));
    }
  }
  template <class T>
  void array(const T* p, size_t count) {
    if (p && count) bytes(p, count * sizeof(T));
    else w.push_back(~0ull);
  }
  bool operator==(const SceneProbe& o) const { return w == o.w; }
};

template <class H>
void walk_scene(const RtcSceneDesc& D, H& h);

Hash128 hash_scene(const RtcSceneDesc& D) {
  Hash128 h;
  walk_scene(D, h);
  return h;
}
SceneProbe probe_scene(const RtcSceneDesc& D) {
  SceneProbe h;
  walk_scene(D, h);
  return h;
}

template <class H>
void walk_scene(const RtcSceneDesc& D, H& h) {
  h.bytes(D.light_position, sizeof D.light_position);
  h.bytes(D.light_intensity, sizeof D.light_intensity);
  const uint32_t counts[6] = {D.n_transforms, D.n_materials, D.n_patterns, D.n_groups, D.n_leaves, D.n_triangles};
  h.bytes(counts, sizeof counts);
  h.array(D.transform_inverse, (size_t)D.n_transforms * 16);
  h.array(D.materials, D.n_materials);
  h.array(D.patterns, D.n_patterns);
  h.array(D.group_parent, D.n_groups);
  h.array(D.group_bounds_min, (size_t)D.n_groups * 3);
  h.array(D.group_bounds_max, (size_t)D.n_groups * 3);
  h.array(D.leaf_kind, D.n_leaves);
  h.array(D.leaf_transform, D.n_leaves);
  h.array(D.leaf_material, D.n_leaves);
  h.array(D.leaf_group, D.n_leaves);
  h.array(D.leaf_class, D.n_leaves);
  h.array(D.leaf_params, (size_t)D.n_leaves * 3);
  h.array(D.leaf_triangle, D.n_leaves);
  h.array(D.tri_p1, (size_t)D.n_triangles * 3);
  h.array(D.tri_e1, (size_t)D.n_triangles * 3);
  h.array(D.tri_e2, (size_t)D.n_triangles * 3);
  h.array(D.tri_normal, (size_t)D.n_triangles * 3);
}

// ---- pinned staging + pipelined copy to pageable memory ---------------------------------------------------
struct Segment {  // `bytes` from device address `src` to host address `dst`
  void* dst;
  const void* src;
  size_t bytes;
};

struct Stager {
  static constexpr size_t PIECE = 4u << 20;
  int device = -1;
  void* pinned = nullptr;
  size_t cap = 0;
  hipStream_t stream = nullptr;
  std::vector<hipEvent_t> events;

  RtcStatus reserve(int dev, size_t bytes, size_t pieces) {
    if (device != dev) {
      release();
      device = dev;
    }
    HIPCHECK(hipSetDevice(dev));
    if (!stream) HIPCHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    if (cap < bytes) {
      if (pinned) (void)hipHostFree(pinned);
      pinned = nullptr, cap = 0;
      HIPCHECK(hipHostMalloc(&pinned, bytes, hipHostMallocDefault));
      cap = bytes;
    }
    while (events.size() < pieces) {
      hipEvent_t e;
      HIPCHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      events.push_back(e);
    }
    return RTC_OK;
  }
  void release() {
    if (device >= 0) (void)hipSetDevice(device);
    if (pinned) (void)hipHostFree(pinned);
    for (hipEvent_t e : events) (void)hipEventDestroy(e);
    if (stream) (void)hipStreamDestroy(stream);
    pinned = nullptr, cap = 0, stream = nullptr, device = -1;
    events.clear();
  }
  // Copies every segment device -> host.  `after` (may be null): an event the device data is complete behind.
  RtcStatus download(int dev, const std::vector<Segment>& segs, hipEvent_t after) {
    struct Piece {
      void* dst;
      const void* src;
      size_t bytes, off;
    };
    std::vector<Piece> pieces;
    size_t total = 0;
    for (const Segment& s : segs)
      for (size_t o = 0; o < s.bytes; o += PIECE) {
        const size_t n = std::min(PIECE, s.bytes - o);
        pieces.push_back({(char*)s.dst + o, (const char*)s.src + o, n, total});
        total += n;
      }
    if (pieces.empty()) return RTC_OK;
    RtcStatus st = reserve(dev, total, pieces.size());
    if (st != RTC_OK) return st;
    if (after) HIPCHECK(hipStreamWaitEvent(stream, after, 0));
    for (size_t i = 0; i < pieces.size(); ++i) {
      HIPCHECK(hipMemcpyAsync((char*)pinned + pieces[i].off, pieces[i].src, pieces[i].bytes, hipMemcpyDeviceToHost, stream));
      HIPCHECK(hipEventRecord(events[i], stream));
    }
    // host side: a few threads take pieces in turn; each waits for its piece's DMA and copies it out
    std::atomic<size_t> next{0};
    std::atomic<int> failed{0};
    auto work = [&]() {
      (void)hipSetDevice(dev);
      for (size_t i = next.fetch_add(1); i < pieces.size(); i = next.fetch_add(1)) {
        if (hipEventSynchronize(events[i]) != hipSuccess) {
          failed = 1;
          continue;
        }
        std::memcpy(pieces[i].dst, (const char*)pinned + pieces[i].off, pieces[i].bytes);
      }
    };
    const size_t helpers = std::min<size_t>(7, pieces.size() - 1);
    std::vector<std::thread> pool;
    for (size_t t = 0; t < helpers; ++t) pool.emplace_back(work);
    work();
    for (std::thread& t : pool) t.join();
    if (failed) return fail(RTC_ERR_HIP, "framebuffer copy failed");
    return RTC_OK;
  }
};

// ---- device framebuffers that outlive a call -------------------------------------------------------------
struct FrameBuffers {
  float* rgb = nullptr;
  double* rgb64 = nullptr;
  size_t cap = 0, cap64 = 0;  // in values
  RtcStatus reserve(size_t n, bool want64) {
    if (cap < n) {
      (void)hipFree(rgb);
      rgb = nullptr, cap = 0;
      HIPCHECK(hipMalloc((void**)&rgb, n * sizeof(float)));
      cap = n;
    }
    if (want64 && cap64 < n) {
      (void)hipFree(rgb64);
      rgb64 = nullptr, cap64 = 0;
      HIPCHECK(hipMalloc((void**)&rgb64, n * sizeof(double)));
      cap64 = n;
    }
    return RTC_OK;
  }
  void release() {
    (void)hipFree(rgb), (void)hipFree(rgb64);
    rgb = nullptr, rgb64 = nullptr, cap = cap64 = 0;
  }
};

// ---- scene cache of rtc_render ---------------------------------------------------------------------------
struct CachedScene {
  uint64_t ha = 0, hb = 0;
  SceneProbe probe;
  int device = 0;
  RtcScene* sc = nullptr;
  FrameBuffers fb;
  hipEvent_t done = nullptr;
  uint64_t tick = 0;
};
constexpr size_t SCENE_CACHE_ENTRIES = 4;  // per device
// Locking.  The scene is hashed before any lock is taken.  rtc_render holds its DEVICE's lock for the whole call (the cached
// scenes of a device, their framebuffers and the device's stager belong to whoever holds it), so calls for different devices
// run concurrently; rtc_render_multi calls hold g_multi_mu (they share replicas and communicators).  g_cache_mu only guards
// the containers themselves and is never held across device work.  Order: device lock / g_multi_mu, then g_cache_mu.
constexpr int MAX_DEVICE_LOCKS = 64;
std::mutex g_device_mu[MAX_DEVICE_LOCKS];
std::mutex g_multi_mu;
std::mutex g_cache_mu;
std::vector<CachedScene*> g_scene_cache;
std::atomic<uint64_t> g_cache_tick{0};
std::vector<Stager*> g_stagers;  // one per device (a cached scene's last launch went to its device's stager stream)
Stager& stager_of(int device) {  // under the device's lock
  std::lock_guard<std::mutex> lock(g_cache_mu);
  if ((size_t)device >= g_stagers.size()) g_stagers.resize((size_t)device + 1, nullptr);
  if (!g_stagers[(size_t)device]) g_stagers[(size_t)device] = new Stager;
  return *g_stagers[(size_t)device];
}

void drop_cached_scene(CachedScene* e) {
  (void)hipSetDevice(e->device);
  e->fb.release();
  if (e->done) (void)hipEventDestroy(e->done);
  rtc_scene_destroy(e->sc);
  delete e;
}

// ---- RCCL, loaded on first use ---------------------------------------------------------------------------
// Only rtc_multi_* needs it, and only with more than one device: a single-GPU host never loads the library.
// The handful of entry points used here are declared by hand (plain C, stable across RCCL 2.x) so that building
// this file does not depend on the RCCL headers.
struct Rccl {
  typedef struct ncclComm* comm_t;
  void* lib = nullptr;
  int (*CommInitAll)(comm_t*, int, const int*) = nullptr;
  int (*CommDestroy)(comm_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Send)(const void*, size_t, int, int, comm_t, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, comm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  static constexpr int kInt8 = 0;  // ncclInt8 = ncclChar = 0: the payload is moved as bytes
  int resolved = 0;
  bool load(std::string* why) {
    if (lib && resolved == 7) return true;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
      lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (lib) break;
    }
    if (!lib) {
      if (why) *why = std::string("cannot load librccl: ") + dlerror();
      return false;
    }
    resolved = 0;
    auto sym = [&](const char* n) {
      void* p = dlsym(lib, n);
      if (p) ++resolved;
      return p;
    };
    CommInitAll = (decltype(CommInitAll))sym("ncclCommInitAll");
    CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
    GroupStart = (decltype(GroupStart))sym("ncclGroupStart");
    GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
    Send = (decltype(Send))sym("ncclSend");
    Recv = (decltype(Recv))sym("ncclRecv");
    GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
    if (resolved != 7) {
      if (why) *why = "librccl lacks one of ncclCommInitAll/CommDestroy/GroupStart/GroupEnd/Send/Recv/GetErrorString";
      return false;
    }
    return true;
  }
};
Rccl g_rccl;

}  // namespace

constexpr uint32_t MULTI_BLOCK_ROWS = 16;  // rows per block dealt to the devices (two pixel tiles of the accelerated kernel)
// A device's blocks are rendered and sent in this many pieces, on streams of alternating priority (high, low): the dispatcher
// prefers the earlier piece's workgroups, so its long-lived waves drain while the next piece already fills the machine and its
// transfer overlaps the rest of the render.  Pieces queued on ONE stream each wait for the previous piece's last wave: measured
// on one MI355X with an eighth of C5's rows (tools/chunk_overlap.py, profiles/r02/v23_*): 1 piece 0.75 ms, 4 pieces on one
// stream 1.92 ms, 2 pieces on a high- and a low-priority stream 0.75 ms with the first piece complete after 0.52 ms.
constexpr uint32_t MULTI_CHUNKS = 2;

struct RtcMulti {
  std::vector<int> devices;          // devices[0] is the root
  std::vector<RtcScene*> scenes;     // one replica per entry of `devices`
  std::vector<hipStream_t> render_stream, render_stream_lo, copy_stream;  // render_stream: high priority, _lo: low
  std::vector<hipEvent_t> chunk_done;  // [device][chunk]
  std::vector<hipEvent_t> t0, t1;      // render time per device
  std::vector<FrameBuffers> local;     // each device's dense row blocks
  std::vector<FrameBuffers> recv;      // on the root: one receive area per peer (entry 0 unused: the root's own `local`)
  hipStream_t root_recv = nullptr;     // on the root: the stream of the RCCL receives (one stream per communicator)
  std::vector<Rccl::comm_t> comms;
  bool peer_enabled = false;
  Stager stager;
  uint64_t ha = 0, hb = 0;  // content hash (rtc_render_multi's cache key)
  SceneProbe probe;         // ... and its second key, compared on a hash match
  uint64_t tick = 0;
};

namespace {
std::vector<RtcMulti*> g_multi_cache;
constexpr size_t MULTI_CACHE_ENTRIES = 2;

// rows of local block j (0-based, in this device's dense buffer) for device r of n: global block r + j*n
uint32_t block_rows_of(uint32_t vsize, uint32_t global_block) {
  const uint32_t y0 = global_block * MULTI_BLOCK_ROWS;
  return y0 >= vsize ? 0u : std::min(MULTI_BLOCK_ROWS, vsize - y0);
}
}  // namespace

// Diagnostics (not part of include/rtc_hip.h): loads librccl and resolves the entry points rtc_multi_render uses;
// returns how many of the 7 were found (no communicator is created, no device is touched).
extern "C" int rtc_debug_rccl_symbols(void) {
  std::string why;
  if (!g_rccl.load(&why)) g_err = why;
  return g_rccl.resolved;
}

void rtc_release_cached(void) {
  // nobody may be inside rtc_render / rtc_render_multi while their caches go away
  std::vector<std::unique_lock<std::mutex>> held;
  for (int i = 0; i < MAX_DEVICE_LOCKS; ++i) held.emplace_back(g_device_mu[i]);
  std::lock_guard<std::mutex> lock_multi(g_multi_mu);
  std::lock_guard<std::mutex> lock(g_cache_mu);
  for (CachedScene* e : g_scene_cache) drop_cached_scene(e);
  g_scene_cache.clear();
  for (RtcMulti* m : g_multi_cache) rtc_multi_destroy(m);
  g_multi_cache.clear();
  for (Stager* st : g_stagers)
    if (st) {
      st->release();
      delete st;
    }
  g_stagers.clear();
}

RtcStatus rtc_render(const RtcSceneDesc* desc, const RtcCameraDesc* cam, const RtcRenderOpts* opts, int device,
                     float* h_rgb, double* h_rgb64, RtcStats* stats) {
  if (!desc || !cam || !opts || (!h_rgb && !h_rgb64)) return fail(RTC_ERR_INVALID_ARG, "null argument");
  if (device < 0 || device >= MAX_DEVICE_LOCKS) return fail(RTC_ERR_NO_DEVICE, "device %d out of range", device);
  const double t_start = now_ms();
  // ---- the scene: resident from an earlier call, or uploaded now -------------------------------------------
  const Hash128 h = hash_scene(*desc);  // (no lock held: calls for other devices are not serialised behind it)
  const SceneProbe probe = probe_scene(*desc);
  std::lock_guard<std::mutex> device_lock(g_device_mu[device]);
  CachedScene* e = nullptr;
  {
    std::lock_guard<std::mutex> lock(g_cache_mu);
    for (CachedScene* c : g_scene_cache)
      if (c->ha == h.a && c->hb == h.b && c->device == device && c->probe == probe) e = c;
  }
  double upload_ms = 0.0;
  if (!e) {
    RtcScene* sc = nullptr;
    RtcStatus st = rtc_scene_create(desc, device, &sc);
    if (st != RTC_OK) return st;
    e = new CachedScene;
    e->ha = h.a, e->hb = h.b, e->probe = probe, e->device = device, e->sc = sc;
    if (hipEventCreateWithFlags(&e->done, hipEventDisableTiming) != hipSuccess) {
      drop_cached_scene(e);
      return fail(RTC_ERR_HIP, "hipEventCreate failed");
    }
    CachedScene* evict = nullptr;
    {
      std::lock_guard<std::mutex> lock(g_cache_mu);
      size_t mine = 0, lru = g_scene_cache.size();
      for (size_t i = 0; i < g_scene_cache.size(); ++i)
        if (g_scene_cache[i]->device == device) {  // only this device's entries: the others may be in use under their own lock
          ++mine;
          if (lru == g_scene_cache.size() || g_scene_cache[i]->tick < g_scene_cache[lru]->tick) lru = i;
        }
      if (mine >= SCENE_CACHE_ENTRIES) {  // evict the least recently used
        evict = g_scene_cache[lru];
        g_scene_cache.erase(g_scene_cache.begin() + (long)lru);
      }
      g_scene_cache.push_back(e);
    }
    if (evict) drop_cached_scene(evict);
    upload_ms = now_ms() - t_start;
  }
  e->tick = ++g_cache_tick;
  RtcScene* sc = e->sc;
  RtcStatus st = validate_camera(sc, cam, opts);
  if (st != RTC_OK) return st;
  const uint32_t rows = rtc_local_rows(cam->vsize, opts);
  const size_t n = (size_t)rows * cam->hsize * 3;
  if (n == 0) {
    if (stats) *stats = RtcStats{};
    return RTC_OK;
  }
  HIPCHECK(hipSetDevice(device));
  st = e->fb.reserve(n, h_rgb64 != nullptr);
  if (st != RTC_OK) return st;
  RtcRenderOpts o = *opts;
  if (stats) o.flags |= RTC_FLAG_TIMING | RTC_FLAG_COUNT_RAYS;  // the work counters cost several times the kernel time: only on request
  Stager& stager = stager_of(device);
  st = stager.reserve(device, 0, 0);  // creates the copy stream the launch goes to
  if (st != RTC_OK) return st;
  double t_dl = 0.0;
  for (;;) {  // once, unless a wave's ray pool overflowed: then again with the pools check_pool_flag has doubled
    if (stats && (st = rtc_scene_reset_counters(sc)) != RTC_OK) return st;
    st = rtc_render_device(sc, cam, &o, e->fb.rgb, h_rgb64 ? e->fb.rgb64 : nullptr, stager.stream);
    if (st != RTC_OK) return st;
    t_dl = now_ms();
    std::vector<Segment> segs;
    if (h_rgb) segs.push_back({h_rgb, e->fb.rgb, n * sizeof(float)});
    if (h_rgb64) segs.push_back({h_rgb64, e->fb.rgb64, n * sizeof(double)});
    st = stager.download(device, segs, nullptr);  // same stream as the kernel: ordered behind it
    if (st != RTC_OK) return st;
    st = check_pool_flag(sc);
    if (st == RTC_ERR_RAY_POOL) continue;
    if (st != RTC_OK) return st;
    break;
  }
  if (stats) {
    RtcStats s{};
    st = rtc_scene_stats(sc, &s);
    if (st != RTC_OK) return st;
    s.upload_ms = upload_ms;
    s.download_ms = now_ms() - t_dl - s.kernel_ms;  // the copy waits for the kernel: keep the two apart
    if (s.download_ms < 0.0) s.download_ms = 0.0;
    *stats = s;
  }
  return RTC_OK;
}

// ---- several devices, one process -------------------------------------------------------------------------
void rtc_multi_destroy(RtcMulti* m) {
  if (!m) return;
  for (size_t i = 0; i < m->devices.size(); ++i) {
    (void)hipSetDevice(m->devices[i]);
    if (i < m->comms.size() && m->comms[i] && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(m->comms[i]);
    if (i < m->local.size()) m->local[i].release();
    if (i < m->render_stream.size() && m->render_stream[i]) (void)hipStreamDestroy(m->render_stream[i]);
    if (i < m->render_stream_lo.size() && m->render_stream_lo[i]) (void)hipStreamDestroy(m->render_stream_lo[i]);
    if (i < m->copy_stream.size() && m->copy_stream[i]) (void)hipStreamDestroy(m->copy_stream[i]);
    if (i < m->t0.size() && m->t0[i]) (void)hipEventDestroy(m->t0[i]);
    if (i < m->t1.size() && m->t1[i]) (void)hipEventDestroy(m->t1[i]);
    for (uint32_t c = 0; c < MULTI_CHUNKS; ++c)
      if (i * MULTI_CHUNKS + c < m->chunk_done.size() && m->chunk_done[i * MULTI_CHUNKS + c])
        (void)hipEventDestroy(m->chunk_done[i * MULTI_CHUNKS + c]);
    if (i < m->scenes.size()) rtc_scene_destroy(m->scenes[i]);
  }
  if (!m->devices.empty()) (void)hipSetDevice(m->devices[0]);
  for (FrameBuffers& f : m->recv) f.release();
  if (m->root_recv) (void)hipStreamDestroy(m->root_recv);
  m->stager.release();
  delete m;
}

RtcStatus rtc_multi_create(const RtcSceneDesc* desc, int n_devices, const int* devices, RtcMulti** out) {
  if (!desc || !out || n_devices < 1 || !devices) return fail(RTC_ERR_INVALID_ARG, "null argument or no devices");
  *out = nullptr;
  const int ndev = rtc_device_count();
  if (ndev <= 0) return fail(RTC_ERR_NO_DEVICE, "no HIP device visible");
  for (int i = 0; i < n_devices; ++i)
    if (devices[i] < 0 || devices[i] >= ndev) return fail(RTC_ERR_NO_DEVICE, "device %d out of range (%d devices)", devices[i], ndev);
  auto* m = new RtcMulti;
  m->devices.assign(devices, devices + n_devices);
  const size_t n = (size_t)n_devices;
  m->scenes.assign(n, nullptr), m->render_stream.assign(n, nullptr), m->render_stream_lo.assign(n, nullptr), m->copy_stream.assign(n, nullptr);
  m->t0.assign(n, nullptr), m->t1.assign(n, nullptr), m->chunk_done.assign(n * MULTI_CHUNKS, nullptr);
  m->local.resize(n), m->recv.resize(n);
  auto bail = [&](RtcStatus s) {
    rtc_multi_destroy(m);
    return s;
  };
  for (size_t i = 0; i < n; ++i) {  // the scene is small (< 1 MB for the reference's meshes): replicate it
    RtcStatus st = rtc_scene_create(desc, m->devices[i], &m->scenes[i]);
    if (st != RTC_OK) return bail(st);
    int pri_least = 0, pri_greatest = 0;  // numerically: greatest priority <= least priority
    if (hipSetDevice(m->devices[i]) != hipSuccess || hipDeviceGetStreamPriorityRange(&pri_least, &pri_greatest) != hipSuccess ||
        hipStreamCreateWithPriority(&m->render_stream[i], hipStreamNonBlocking, pri_greatest) != hipSuccess ||
        hipStreamCreateWithPriority(&m->render_stream_lo[i], hipStreamNonBlocking, pri_least) != hipSuccess ||
        hipStreamCreateWithFlags(&m->copy_stream[i], hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&m->t0[i]) != hipSuccess || hipEventCreate(&m->t1[i]) != hipSuccess)
      return bail(fail(RTC_ERR_HIP, "stream / event creation failed on device %d", m->devices[i]));
    for (uint32_t c = 0; c < MULTI_CHUNKS; ++c)
      if (hipEventCreateWithFlags(&m->chunk_done[i * MULTI_CHUNKS + c], hipEventDisableTiming) != hipSuccess)
        return bail(fail(RTC_ERR_HIP, "hipEventCreate failed"));
  }
  if (hipSetDevice(m->devices[0]) != hipSuccess) return bail(fail(RTC_ERR_HIP, "hipSetDevice failed"));
  if (hipStreamCreateWithFlags(&m->root_recv, hipStreamNonBlocking) != hipSuccess)
    return bail(fail(RTC_ERR_HIP, "stream creation failed on the root device"));
  const Hash128 h = hash_scene(*desc);
  m->ha = h.a, m->hb = h.b;
  *out = m;
  return RTC_OK;
}

static RtcStatus multi_render_once(RtcMulti* m, const RtcCameraDesc* cam, const RtcRenderOpts* opts, float* h_rgb, double* h_rgb64,
                                   RtcStats* stats);
RtcStatus rtc_multi_render(RtcMulti* m, const RtcCameraDesc* cam, const RtcRenderOpts* opts, float* h_rgb, double* h_rgb64,
                           RtcStats* stats) {
  for (;;) {  // once, unless a wave's ray pool overflowed on some device: then again with that replica's pools doubled
    const RtcStatus st = multi_render_once(m, cam, opts, h_rgb, h_rgb64, stats);
    if (st != RTC_ERR_RAY_POOL) return st;
  }
}
static RtcStatus multi_render_once(RtcMulti* m, const RtcCameraDesc* cam, const RtcRenderOpts* opts, float* h_rgb, double* h_rgb64,
                                   RtcStats* stats) {
  if (!m || !cam || !opts || (!h_rgb && !h_rgb64)) return fail(RTC_ERR_INVALID_ARG, "null argument");
  if (opts->block_rows || opts->block_first || opts->block_step)
    return fail(RTC_ERR_INVALID_ARG, "rtc_multi_render deals the row blocks itself: block_rows/first/step must be 0");
  const uint32_t n = (uint32_t)m->devices.size();
  const bool p2p = (opts->flags & RTC_FLAG_GATHER_P2P) != 0;
  RtcStatus st = validate_camera(m->scenes[0], cam, opts);
  if (st != RTC_OK) return st;
  if (!p2p && n > 1) {
    for (uint32_t i = 0; i < n; ++i)
      for (uint32_t j = 0; j < i; ++j)
        if (m->devices[i] == m->devices[j])
          return fail(RTC_ERR_INVALID_ARG, "device %d listed twice: RCCL needs distinct devices (RTC_FLAG_GATHER_P2P accepts it)", m->devices[i]);
  }
  const uint32_t W = cam->hsize, H = cam->vsize;
  const uint32_t nblocks = (H + MULTI_BLOCK_ROWS - 1) / MULTI_BLOCK_ROWS;
  const uint32_t nb_max = (nblocks + n - 1) / n;  // blocks of the largest shard
  const size_t row_vals = (size_t)W * 3;
  const size_t shard_vals = (size_t)nb_max * MULTI_BLOCK_ROWS * row_vals;
  const bool want64 = h_rgb64 != nullptr;
  const uint32_t flags = (opts->flags & (RTC_FLAG_EXACT_SWEEP)) | (stats ? (uint32_t)RTC_FLAG_COUNT_RAYS : 0u);
  const double t_start = now_ms();

  // ---- transport set-up (first frame only) ---------------------------------------------------------------
  if (n > 1 && p2p && !m->peer_enabled) {
    for (uint32_t i = 1; i < n; ++i) {
      if (m->devices[i] == m->devices[0]) continue;
      int can = 0;
      HIPCHECK(hipDeviceCanAccessPeer(&can, m->devices[i], m->devices[0]));
      if (!can) return fail(RTC_ERR_HIP, "device %d cannot access device %d", m->devices[i], m->devices[0]);
      HIPCHECK(hipSetDevice(m->devices[i]));
      const hipError_t pe = hipDeviceEnablePeerAccess(m->devices[0], 0);
      if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) return fail(RTC_ERR_HIP, "hipDeviceEnablePeerAccess failed");
      (void)hipGetLastError();
    }
    m->peer_enabled = true;
  }
  if (n > 1 && !p2p && m->comms.empty()) {
    std::string why;
    if (!g_rccl.load(&why)) return fail(RTC_ERR_COLLECTIVE, "%s", why.c_str());
    m->comms.assign(n, nullptr);
    const int rc = g_rccl.CommInitAll(m->comms.data(), (int)n, m->devices.data());
    if (rc != 0) {
      m->comms.clear();
      return fail(RTC_ERR_COLLECTIVE, "ncclCommInitAll: %s", g_rccl.GetErrorString(rc));
    }
  }
  // ---- buffers -----------------------------------------------------------------------------------------
  for (uint32_t i = 0; i < n; ++i) {
    HIPCHECK(hipSetDevice(m->devices[i]));
    st = m->local[i].reserve(shard_vals, want64);
    if (st != RTC_OK) return st;
    if (stats && (st = rtc_scene_reset_counters(m->scenes[i])) != RTC_OK) return st;
  }
  HIPCHECK(hipSetDevice(m->devices[0]));
  for (uint32_t i = 1; i < n; ++i)
    if ((st = m->recv[i].reserve(shard_vals, want64)) != RTC_OK) return st;

  // ---- render in chunks; a finished chunk leaves for the root while the next one renders ---------------------
  // device i owns global blocks i, i+n, ...: local block j <-> global block i + j*n
  const uint32_t chunks = n == 1 ? 1u : MULTI_CHUNKS;  // nothing to overlap on one device: one launch, one tail
  auto chunk_range = [&](uint32_t nb, uint32_t c, uint32_t& j0, uint32_t& j1) {
    j0 = (uint32_t)((uint64_t)nb * c / chunks), j1 = (uint32_t)((uint64_t)nb * (c + 1) / chunks);
  };
  std::vector<uint32_t> nb(n);
  for (uint32_t i = 0; i < n; ++i) nb[i] = i < nblocks ? (nblocks - i + n - 1) / n : 0u;
  // From here on work is in flight on several devices.  An error must not leave it there (the next call would reuse
  // buffers that are still being written) nor leave an RCCL group open (the next call's sends would queue into it):
  // every failing exit below goes through abandon(), which closes the group if one is open and waits for every stream
  // this call has touched.  The first error is the one reported.
  bool group_open = false;
  auto abandon = [&](RtcStatus s) {
    const std::string first = g_err;
    if (group_open) (void)g_rccl.GroupEnd(), group_open = false;
    for (uint32_t i = 0; i < n; ++i) {
      if (hipSetDevice(m->devices[i]) != hipSuccess) continue;
      (void)hipStreamSynchronize(m->render_stream[i]), (void)hipStreamSynchronize(m->render_stream_lo[i]);
      (void)hipStreamSynchronize(m->copy_stream[i]);
    }
    if (hipSetDevice(m->devices[0]) == hipSuccess && m->root_recv) (void)hipStreamSynchronize(m->root_recv);
    (void)hipGetLastError();
    g_err = first;
    return s;
  };
#define MULTI_HIPCHECK(expr)                                                                                      \
  do {                                                                                                            \
    hipError_t e_ = (expr);                                                                                       \
    if (e_ != hipSuccess) return abandon(fail(RTC_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_))); \
  } while (0)
  for (uint32_t i = 0; i < n; ++i) {
    MULTI_HIPCHECK(hipSetDevice(m->devices[i]));
    MULTI_HIPCHECK(hipEventRecord(m->t0[i], m->render_stream[i]));
  }
  struct Msg {  // one peer's piece of this chunk
    uint32_t peer;
    size_t off, vals;
  };
  std::vector<Msg> msgs;
  for (uint32_t c = 0; c < chunks; ++c) {
    for (uint32_t i = 0; i < n; ++i) {
      uint32_t j0, j1;
      chunk_range(nb[i], c, j0, j1);
      MULTI_HIPCHECK(hipSetDevice(m->devices[i]));
      const hipStream_t rs = (c & 1u) ? m->render_stream_lo[i] : m->render_stream[i];
      if (j1 > j0) {
        RowMap R;
        R.block_rows = MULTI_BLOCK_ROWS, R.block_first = i + j0 * n, R.block_step = n;
        R.local_rows = 0, R.tile_row0 = 0, R.tiles_x = R.tiles_y = 0, R.rx0 = R.ry0 = R.rx1 = R.ry1 = R.n_first = 0, R.n_order = 0;
        for (uint32_t j = j0; j < j1; ++j) R.local_rows += block_rows_of(H, i + j * n);
        const size_t off = (size_t)j0 * MULTI_BLOCK_ROWS * row_vals;
        st = launch_render(m->scenes[i], cam, R, (int)opts->recursion_limit, flags, m->local[i].rgb + off,
                           want64 ? m->local[i].rgb64 + off : nullptr, rs);
        if (st != RTC_OK) return abandon(st);
      }
      MULTI_HIPCHECK(hipEventRecord(m->chunk_done[i * MULTI_CHUNKS + c], rs));
      if (c + 1 == chunks) {  // the device's render ends when both streams are done: whoever waits for render_stream waits for both
        if (rs != m->render_stream[i]) MULTI_HIPCHECK(hipStreamWaitEvent(m->render_stream[i], m->chunk_done[i * MULTI_CHUNKS + c], 0));
        MULTI_HIPCHECK(hipEventRecord(m->t1[i], m->render_stream[i]));
      }
    }
    if (n == 1) continue;
    // the exchange step of the path: peers -> root, one message per peer and chunk (f32, then f64 if wanted).
    // First every HIP call that can fail (the copy streams wait for their chunk), then -- RCCL transport -- a group that
    // contains nothing but ncclSend / ncclRecv, so that no early return can sit between ncclGroupStart and ncclGroupEnd.
    msgs.clear();
    for (uint32_t i = 1; i < n; ++i) {
      uint32_t j0, j1;
      chunk_range(nb[i], c, j0, j1);
      if (j1 <= j0) continue;
      Msg g{i, (size_t)j0 * MULTI_BLOCK_ROWS * row_vals, 0};
      for (uint32_t j = j0; j < j1; ++j) g.vals += (size_t)block_rows_of(H, i + j * n) * row_vals;
      MULTI_HIPCHECK(hipSetDevice(m->devices[i]));
      MULTI_HIPCHECK(hipStreamWaitEvent(m->copy_stream[i], m->chunk_done[i * MULTI_CHUNKS + c], 0));
      msgs.push_back(g);
    }
    if (p2p) {
      for (const Msg& g : msgs) {
        const uint32_t i = g.peer;
        MULTI_HIPCHECK(hipSetDevice(m->devices[i]));
        MULTI_HIPCHECK(hipMemcpyPeerAsync(m->recv[i].rgb + g.off, m->devices[0], m->local[i].rgb + g.off, m->devices[i],
                                          g.vals * sizeof(float), m->copy_stream[i]));
        if (want64)
          MULTI_HIPCHECK(hipMemcpyPeerAsync(m->recv[i].rgb64 + g.off, m->devices[0], m->local[i].rgb64 + g.off, m->devices[i],
                                            g.vals * sizeof(double), m->copy_stream[i]));
      }
    } else if (!msgs.empty()) {
      int rc = g_rccl.GroupStart();
      if (rc != 0) return abandon(fail(RTC_ERR_COLLECTIVE, "ncclGroupStart: %s", g_rccl.GetErrorString(rc)));
      group_open = true;
      for (const Msg& g : msgs) {
        const uint32_t i = g.peer;
        rc = g_rccl.Send(m->local[i].rgb + g.off, g.vals * sizeof(float), Rccl::kInt8, 0, m->comms[i], m->copy_stream[i]);
        if (rc == 0) rc = g_rccl.Recv(m->recv[i].rgb + g.off, g.vals * sizeof(float), Rccl::kInt8, (int)i, m->comms[0], m->root_recv);
        if (rc == 0 && want64) {
          rc = g_rccl.Send(m->local[i].rgb64 + g.off, g.vals * sizeof(double), Rccl::kInt8, 0, m->comms[i], m->copy_stream[i]);
          if (rc == 0) rc = g_rccl.Recv(m->recv[i].rgb64 + g.off, g.vals * sizeof(double), Rccl::kInt8, (int)i, m->comms[0], m->root_recv);
        }
        if (rc != 0) return abandon(fail(RTC_ERR_COLLECTIVE, "ncclSend/ncclRecv: %s", g_rccl.GetErrorString(rc)));
      }
      group_open = false;
      rc = g_rccl.GroupEnd();
      if (rc != 0) return abandon(fail(RTC_ERR_COLLECTIVE, "ncclGroupEnd: %s", g_rccl.GetErrorString(rc)));
    }
  }
  // ---- the root waits for every peer's data, then the Canvas goes to the host block by block -----------------
  const double t_gather = now_ms();
  for (uint32_t i = 1; i < n; ++i) {
    MULTI_HIPCHECK(hipSetDevice(m->devices[i]));
    MULTI_HIPCHECK(hipStreamSynchronize(m->copy_stream[i]));
  }
  MULTI_HIPCHECK(hipSetDevice(m->devices[0]));
  if (n > 1 && !p2p) MULTI_HIPCHECK(hipStreamSynchronize(m->root_recv));
  MULTI_HIPCHECK(hipStreamSynchronize(m->render_stream[0]));
  std::vector<Segment> segs;
  segs.reserve((size_t)nblocks * 2);
  for (int pass = 0; pass < 2; ++pass) {  // Canvas order: global block b lives in device b % n's local block b / n
    if ((pass == 0 && !h_rgb) || (pass == 1 && !h_rgb64)) continue;
    for (uint32_t b = 0; b < nblocks; ++b) {
      const uint32_t i = b % n, j = b / n;
      const size_t off = (size_t)j * MULTI_BLOCK_ROWS * row_vals, dst = (size_t)b * MULTI_BLOCK_ROWS * row_vals;
      const size_t vals = (size_t)block_rows_of(H, b) * row_vals;
      if (pass == 0)
        segs.push_back({h_rgb + dst, (i == 0 ? m->local[0].rgb : m->recv[i].rgb) + off, vals * sizeof(float)});
      else
        segs.push_back({h_rgb64 + dst, (i == 0 ? m->local[0].rgb64 : m->recv[i].rgb64) + off, vals * sizeof(double)});
    }
  }
  st = m->stager.download(m->devices[0], segs, nullptr);
  if (st != RTC_OK) return abandon(st);
  for (uint32_t i = 0; i < n; ++i) {
    MULTI_HIPCHECK(hipSetDevice(m->devices[i]));
    MULTI_HIPCHECK(hipStreamSynchronize(m->render_stream[i]));
    if ((st = check_pool_flag(m->scenes[i])) != RTC_OK) return abandon(st);
  }
  if (stats) {
    RtcStats total{};
    for (uint32_t i = 0; i < n; ++i) {
      MULTI_HIPCHECK(hipSetDevice(m->devices[i]));
      MULTI_HIPCHECK(hipStreamSynchronize(m->render_stream[i]));
      unsigned long long c[RTC_N_COUNTERS] = {0};
      MULTI_HIPCHECK(hipMemcpy(c, m->scenes[i]->d_counters, sizeof c, hipMemcpyDeviceToHost));
      total.radiance_rays += c[0], total.shadow_rays += c[1], total.triangle_tests += c[2], total.primitive_tests += c[3];
      total.filter_tests += c[4], total.box_tests += c[5], total.culled_rays += c[20], total.unswept_shadow_rays += c[26];
      float ms = 0.f;
      MULTI_HIPCHECK(hipEventElapsedTime(&ms, m->t0[i], m->t1[i]));
      total.kernel_ms = std::max(total.kernel_ms, (double)ms);
    }
    total.upload_ms = 0.0;
    total.download_ms = now_ms() - t_gather;
    (void)t_start;
    *stats = total;
  }
  return RTC_OK;
#undef MULTI_HIPCHECK
}

RtcStatus rtc_render_multi(const RtcSceneDesc* desc, const RtcCameraDesc* cam, const RtcRenderOpts* opts, int n_devices,
                           const int* devices, float* h_rgb, double* h_rgb64, RtcStats* stats) {
  if (!desc || !cam || !opts || n_devices < 1 || !devices) return fail(RTC_ERR_INVALID_ARG, "null argument or no devices");
  const double t0 = now_ms();
  const Hash128 h = hash_scene(*desc);
  const SceneProbe probe = probe_scene(*desc);
  std::lock_guard<std::mutex> lock(g_multi_mu);
  RtcMulti* m = nullptr;
  for (RtcMulti* c : g_multi_cache)
    if (c->ha == h.a && c->hb == h.b && c->probe == probe && c->devices.size() == (size_t)n_devices &&
        std::equal(c->devices.begin(), c->devices.end(), devices))
      m = c;
  double upload_ms = 0.0;
  if (!m) {
    RtcStatus st = rtc_multi_create(desc, n_devices, devices, &m);
    if (st != RTC_OK) return st;
    m->probe = probe;
    if (g_multi_cache.size() >= MULTI_CACHE_ENTRIES) {
      size_t lru = 0;
      for (size_t i = 1; i < g_multi_cache.size(); ++i)
        if (g_multi_cache[i]->tick < g_multi_cache[lru]->tick) lru = i;
      rtc_multi_destroy(g_multi_cache[lru]);
      g_multi_cache.erase(g_multi_cache.begin() + (long)lru);
    }
    g_multi_cache.push_back(m);
    upload_ms = now_ms() - t0;
  }
  m->tick = ++g_cache_tick;
  RtcStatus st = rtc_multi_render(m, cam, opts, h_rgb, h_rgb64, stats);
  if (st == RTC_OK && stats) stats->upload_ms = upload_ms;
  return st;
}
