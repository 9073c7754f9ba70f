// How fast does the hardware start the workgroups of one launch?  Every workgroup stamps the constant 100 MHz
// clock when its first wave begins, then spins for a fixed time.  Prints, per XCD (block index % 8), the median
// start time relative to the first workgroup, for kernels that differ in kernarg size, register use and LDS.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/dispatch_probe.hip -o tools/micro/dispatch_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

struct Big {
    float v[96]; // 384 bytes of kernel arguments
};

template <int LDS_BYTES, int VGPRS>
__device__ __forceinline__ void body(unsigned long long* start, unsigned spin_ticks, int extra) {
    __shared__ unsigned char lds[LDS_BYTES > 0 ? LDS_BYTES : 1];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (extra) lds[threadIdx.x] = (unsigned char)threadIdx.x; // keeps the allocation alive, never taken
    if (VGPRS >= 96) asm volatile("v_mov_b32 v95, 0" ::: "v95"); // raises the kernel's register allocation
    if (threadIdx.x == 0) start[blockIdx.x] = t0;
    while (__builtin_amdgcn_s_memrealtime() - t0 < spin_ticks) __builtin_amdgcn_s_sleep(2);
    if (extra) start[blockIdx.x] += lds[(threadIdx.x * 7) & 255];
}

template <int LDS_BYTES, int VGPRS>
__global__ __launch_bounds__(256) void probe(unsigned long long* start, unsigned spin_ticks, int extra) {
    body<LDS_BYTES, VGPRS>(start, spin_ticks, extra);
}

template <int LDS_BYTES, int VGPRS>
__global__ __launch_bounds__(256) void probe_big(Big b, unsigned long long* start, unsigned spin_ticks, int extra) {
    body<LDS_BYTES, VGPRS>(start, spin_ticks, extra);
    if (extra) start[blockIdx.x] += (unsigned long long)b.v[threadIdx.x % 96];
}

static void report(const char* name, int blocks, const std::vector<unsigned long long>& h) {
    const unsigned long long first = *std::min_element(h.begin(), h.end());
    std::printf("%-34s blocks %5d: median start per XCD (us):", name, blocks);
    double worst = 0;
    for (int x = 0; x < 8; x++) {
        std::vector<double> s;
        for (int i = x; i < blocks; i += 8) s.push_back((double)(h[i] - first) / 100.0);
        std::sort(s.begin(), s.end());
        std::printf(" %.2f", s[s.size() / 2]);
        worst = std::max(worst, s.back());
    }
    std::printf(" | last start %.2f\n", worst);
}

template <class Launch>
static void run(const char* name, int blocks, Launch launch, bool back_to_back) {
    unsigned long long* d;
    hipMalloc(&d, blocks * sizeof(unsigned long long));
    std::vector<unsigned long long> h(blocks);
    for (int rep = 0; rep < 4; rep++) {
        launch(d);
        if (!back_to_back) hipDeviceSynchronize();
    }
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    report(name, blocks, h);
    hipFree(d);
}

int main() {
    const unsigned spin = 300; // 3 us
    Big big = {};
    for (int b2b = 0; b2b < 2; b2b++) {
        std::printf(b2b ? "--- launches back to back\n" : "--- device idle before the launch\n");
        for (int blocks : {812, 2837}) {
            run("small args", blocks, [&](unsigned long long* d) { hipLaunchKernelGGL((probe<0, 0>), dim3(blocks), dim3(256), 0, 0, d, spin, 0); }, b2b);
            run("small args, 16 KiB LDS", blocks, [&](unsigned long long* d) { hipLaunchKernelGGL((probe<16384, 0>), dim3(blocks), dim3(256), 0, 0, d, spin, 0); }, b2b);
            run("small args, 96 VGPRs", blocks, [&](unsigned long long* d) { hipLaunchKernelGGL((probe<0, 96>), dim3(blocks), dim3(256), 0, 0, d, spin, 0); }, b2b);
            run("384 B args", blocks, [&](unsigned long long* d) { hipLaunchKernelGGL((probe_big<0, 0>), dim3(blocks), dim3(256), 0, 0, big, d, spin, 0); }, b2b);
            run("384 B args, 16 KiB LDS, 96 VGPRs", blocks, [&](unsigned long long* d) { hipLaunchKernelGGL((probe_big<16384, 96>), dim3(blocks), dim3(256), 0, 0, big, d, spin, 0); }, b2b);
        }
    }
    return 0;
}
