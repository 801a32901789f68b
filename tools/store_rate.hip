// Microbenchmark (round 3): what the chip sustains on the WRITE STREAM of the cost volume alone -- the [C, D, h, w] fp32 volume
// of BASELINE config 2 (15.69 GB), no arithmetic -- as a function of the SHAPE in which workgroups emit it:
//   * patch of reference pixels a workgroup owns (PWxPH: 32x4 is the sweep kernel's; 64x2, 128x1, 64x4, 128x2, 464x1),
//   * order in which patches are handed to the XCDs (vertical neighbours first = the sweep kernel's, or horizontal first),
//   * whether the 8 waves of a workgroup take interleaved planes (d = sub + 4 j, the sweep's) or consecutive ones.
// A wave-instruction is one `global_store_dword ... nt` per lane: 64 lanes cover PW-wide row segments of the patch.
// This is the floor any kernel producing that volume has, whatever its arithmetic costs; hipMemsetAsync of the same bytes
// is printed beside it.
//   hipcc --offload-arch=gfx950 -O3 tools/store_rate.hip -o tools/store_rate
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int C = 32, D = 384, H = 688, W = 464;

// workgroup = PW x PH pixels x 128 planes x 32 channels, 8 waves.  The patch has PW*PH/64 "pixel waves"; the 8 waves are
// (pixel wave, depth sub-range) pairs: NPW = PW*PH/64, NSUB = 8 / NPW.
template <int PW, int PH, int XFIRST, int CONSEC>
__global__ __launch_bounds__(512) void store_kernel(float* out, int tiles_x, int tiles_y, int nseg) {
    constexpr int NPW = PW * PH / 64, NSUB = 8 / NPW;
    static_assert(NPW >= 1 && NPW <= 8 && 8 % NPW == 0, "patch must be 1, 2, 4 or 8 waves");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int b = blockIdx.x;
    const int per = gridDim.x / 8;
    if (gridDim.x % 8 == 0) b = (b % 8) * per + b / 8;
    int tx, ty, seg;
    if (XFIRST) { tx = b % tiles_x; seg = (b / tiles_x) % nseg; ty = b / (tiles_x * nseg); }
    else { ty = b % tiles_y; seg = (b / tiles_y) % nseg; tx = b / (tiles_y * nseg); }
    const int pw = wave % NPW, sub = wave / NPW;
    const size_t plane = (size_t)H * W;
    // lanes of a pixel wave: row-major over the patch, 64 consecutive pixels
    const int pl = pw * 64 + lane;
    const int px = tx * PW + pl % PW, py = ty * PH + pl / PW;
    if (px >= W || py >= H) return;
    const unsigned off = (unsigned)(py * W + px) * 4u;
    const float v = (float)lane;
    const int per_wave = 128 / NSUB;
    const unsigned long long cs = (unsigned long long)D * plane * 4;
    for (int g = 0; g < 2; ++g)
        for (int j = 0; j < per_wave; ++j) {
            const int d = seg * 128 + (CONSEC ? sub * per_wave + j : sub + NSUB * j);
            unsigned long long sb = (unsigned long long)(out + ((size_t)(g * 16) * D + d) * plane);
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)sb), hi = __builtin_amdgcn_readfirstlane((unsigned)(sb >> 32));
            sb = ((unsigned long long)hi << 32) | lo;
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                asm volatile("global_store_dword %0, %1, %2 nt" : : "v"(off), "v"(v), "s"(sb));
                sb += cs;
            }
        }
}

template <int PW, int PH, int XFIRST, int CONSEC>
static void run(float* out) {
    const int tiles_x = (W + PW - 1) / PW, tiles_y = (H + PH - 1) / PH, nseg = D / 128;
    const int nblk = tiles_x * tiles_y * nseg;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9f, sum = 0;
    const int reps = 5;
    for (int r = 0; r < reps; ++r) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((store_kernel<PW, PH, XFIRST, CONSEC>), dim3(nblk), dim3(512), 0, 0, out, tiles_x, tiles_y, nseg);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (r > 0) { sum += ms; if (ms < best) best = ms; }
    }
    const double gb = (double)C * D * H * W * 4 / 1e9;
    printf("  patch %3d x %d  %-16s %-18s mean %.3f ms (best %.3f)  %.0f GB/s\n", PW, PH, XFIRST ? "horizontal first" : "vertical first",
           CONSEC ? "consecutive planes" : "interleaved planes", sum / (reps - 1), best, gb / (sum / (reps - 1)) * 1e3);
}

#define ALL(PW, PH) run<PW, PH, 0, 0>(out); run<PW, PH, 1, 0>(out); run<PW, PH, 0, 1>(out); run<PW, PH, 1, 1>(out);

int main() {
    float* out;
    const size_t bytes = (size_t)C * D * H * W * 4;
    if (hipMalloc(&out, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    printf("write stream of the [32,384,688,464] fp32 volume alone (%.2f GB), 4-byte nt stores:\n", bytes / 1e9);
    ALL(32, 4) ALL(64, 2) ALL(128, 1) ALL(64, 4) ALL(128, 2) ALL(32, 8) ALL(16, 4) ALL(64, 8) ALL(128, 4)
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipMemsetAsync(out, 0, bytes, 0);
    (void)hipEventRecord(e0);
    (void)hipMemsetAsync(out, 0, bytes, 0);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("  %-46s %.3f ms  %.0f GB/s\n", "hipMemsetAsync of the same bytes", ms, bytes / 1e9 / ms * 1e3);
    return 0;
}
