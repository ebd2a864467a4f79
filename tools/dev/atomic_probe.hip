// How fast are fp32 atomic adds, as a function of the address pattern?  (gather_bwd's scatter: 38 M atomics in 1.5 ms.)
//   0: every lane its own random float (the reference-layout scatter: 9 planes, one instruction per channel)
//   1: groups of 9 consecutive lanes add to 9 consecutive floats of a random 48-byte record (channel-last record per corner,
//      lanes = channels: ONE instruction per 7 records)
//   2: like 0 but the 9 channels of a record issued as 9 instructions to consecutive floats (the round-3 experiment)
//   3: groups of 12 lanes -> one 48-byte record, 16-byte aligned
// hipcc --offload-arch=gfx950 -O3 tools/dev/atomic_probe.hip -o /tmp/atomic_probe && /tmp/atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int MODE>
__global__ void probe(float* buf, unsigned n_rec, int per_thread) {
  const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned lane = threadIdx.x & 63, wave = tid >> 6;
  for (int it = 0; it < per_thread; ++it) {
    if (MODE == 0) {
      const unsigned r = hash(tid * 977u + it) % (n_rec * 12u);
      unsafeAtomicAdd(buf + r, 1.f);
    } else if (MODE == 1) {
      const unsigned grp = lane / 9, c = lane % 9;
      if (grp < 7) {
        const unsigned r = hash((wave * 7u + grp) * 131u + it) % n_rec;
        unsafeAtomicAdd(buf + (size_t)r * 12 + c, 1.f);
      }
    } else if (MODE == 2) {
      const unsigned r = hash(tid * 977u + it) % n_rec;
#pragma unroll
      for (int c = 0; c < 9; ++c) unsafeAtomicAdd(buf + (size_t)r * 12 + c, 1.f);
    } else {
      const unsigned grp = lane / 12, c = lane % 12;
      if (grp < 5) {
        const unsigned r = hash((wave * 5u + grp) * 131u + it) % n_rec;
        unsafeAtomicAdd(buf + (size_t)r * 12 + c, 1.f);
      }
    }
  }
}

template <int MODE>
double run(float* buf, unsigned n_rec, int blocks, int per_thread) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  probe<MODE><<<blocks, 256>>>(buf, n_rec, per_thread);
  hipDeviceSynchronize();
  hipEventRecord(a);
  probe<MODE><<<blocks, 256>>>(buf, n_rec, per_thread);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return ms;
}

int main() {
  const unsigned n_rec = 14u << 20;    // 14 M records x 48 B = 672 MB (the gradient volumes' size)
  float* buf;
  hipMalloc(&buf, (size_t)n_rec * 48);
  hipMemset(buf, 0, (size_t)n_rec * 48);
  const int blocks = 2048, per = 16;
  const double threads = blocks * 256.0;
  double t0 = run<0>(buf, n_rec, blocks, per * 9);
  printf("mode 0 (scattered floats):        %.3f ms for %.1f M atomics = %.1f G/s\n", t0, threads * per * 9 / 1e6, threads * per * 9 / t0 / 1e6);
  double t1 = run<1>(buf, n_rec, blocks, per * 9);
  printf("mode 1 (9 lanes -> 36 B record):   %.3f ms for %.1f M atomics (%.1f M records) = %.1f G atomics/s, %.1f G records/s\n", t1,
         threads / 64 * 63 * per * 9 / 1e6, threads / 64 * 7 * per * 9 / 1e6, threads / 64 * 63 * per * 9 / t1 / 1e6, threads / 64 * 7 * per * 9 / t1 / 1e6);
  double t2 = run<2>(buf, n_rec, blocks, per);
  printf("mode 2 (9 instructions / record):  %.3f ms for %.1f M atomics = %.1f G/s\n", t2, threads * per * 9 / 1e6, threads * per * 9 / t2 / 1e6);
  double t3 = run<3>(buf, n_rec, blocks, per * 9);
  printf("mode 3 (12 lanes -> 48 B record):  %.3f ms for %.1f M atomics (%.1f M records) = %.1f G atomics/s, %.1f G records/s\n", t3,
         threads / 64 * 60 * per * 9 / 1e6, threads / 64 * 5 * per * 9 / 1e6, threads / 64 * 60 * per * 9 / t3 / 1e6, threads / 64 * 5 * per * 9 / t3 / 1e6);
  return 0;
}
