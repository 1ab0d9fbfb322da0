// Dispatch floor of one-wave workgroups on MI355X: how long does a launch of N tiny workgroups take when each
// only writes its 8x8x3 floats, with / without an LDS allocation and a scratch (private memory) reservation that is never touched?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/dispatch_floor tools/micro/dispatch_floor.hip && /tmp/dispatch_floor
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x)                                                              \
  do {                                                                        \
    hipError_t e = (x);                                                       \
    if (e != hipSuccess) {                                                    \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); \
      exit(1);                                                                \
    }                                                                         \
  } while (0)

template <int LDS_BYTES, int SCRATCH_WORDS, int THREADS>
__global__ __launch_bounds__(THREADS) void tiny(float* out, int w, int h, int seed) {
  __shared__ float lds[LDS_BYTES ? LDS_BYTES / 4 : 1];
  float priv[SCRATCH_WORDS ? SCRATCH_WORDS : 1];
  const int side = THREADS == 64 ? 8 : 16;
  const int tx = threadIdx.x % side, ty = threadIdx.x / side;
  const int px = blockIdx.x * side + tx, py = blockIdx.y * side + ty;
  float v = 0.f;
  if (SCRATCH_WORDS && seed == 12345) {  // reserved (the private segment is sized for it) but never touched at run time
    for (int i = 0; i < SCRATCH_WORDS; ++i) priv[i] = (float)(i + seed);
    v += priv[(px + seed) % SCRATCH_WORDS];  // dynamic index keeps the array in scratch
  }
  if (LDS_BYTES) {
    lds[threadIdx.x] = v;
    v = lds[(threadIdx.x + seed) % THREADS];
  }
  if (px < w && py < h) {
    const size_t i = ((size_t)py * w + px) * 3;
    out[i] = v, out[i + 1] = v, out[i + 2] = v;
  }
}

template <class K>
static void run(const char* name, K kernel, dim3 grid, dim3 block, float* out, int w, int h) {
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kernel, grid, block, 0, 0, out, w, h, 0);
  CHECK(hipDeviceSynchronize());
  const int reps = 20;
  CHECK(hipEventRecord(a, 0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kernel, grid, block, 0, 0, out, w, h, 0);
  CHECK(hipEventRecord(b, 0));
  CHECK(hipEventSynchronize(b));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, a, b));
  printf("%-44s grid %6u x %3u threads: %8.2f us per launch\n", name, grid.x * grid.y, block.x, ms / reps * 1e3);
}

int main() {
  const int w = 1920, h = 1080;
  float* out;
  CHECK(hipMalloc(&out, (size_t)w * h * 3 * sizeof(float)));
  run("64 thr, no LDS, no scratch", tiny<0, 0, 64>, dim3(w / 8, h / 8), dim3(64), out, w, h);
  run("64 thr, 8 KiB LDS", tiny<8192, 0, 64>, dim3(w / 8, h / 8), dim3(64), out, w, h);
  run("64 thr, 704 B scratch/lane", tiny<0, 176, 64>, dim3(w / 8, h / 8), dim3(64), out, w, h);
  run("64 thr, 8 KiB LDS + 704 B scratch", tiny<8192, 176, 64>, dim3(w / 8, h / 8), dim3(64), out, w, h);
  run("256 thr, no LDS, no scratch", tiny<0, 0, 256>, dim3(w / 16, (h + 15) / 16), dim3(256), out, w, h);
  run("256 thr, 8 KiB LDS + 704 B scratch", tiny<8192, 176, 256>, dim3(w / 16, (h + 15) / 16), dim3(256), out, w, h);
  run("64 thr, no LDS/scratch, 4x fewer WGs", tiny<0, 0, 64>, dim3(w / 8, h / 32), dim3(64), out, w, h);
  CHECK(hipFree(out));
  return 0;
}
