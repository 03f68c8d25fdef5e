// What the board lets the matrix pipe do (development tool): a kernel that does NOTHING but `v_mfma_f32_16x16x32_bf16` from registers (no LDS, no memory), looped for a few
// seconds while a host thread reads the amdgpu hwmon power / clock of this GPU.  The 2.5 PFLOP/s dense bf16 peak is the matrix pipes at 2.4 GHz; the board's 1400 W cap holds a
// kernel like this one (and the engine's 3x3 convolutions) below that clock -- this probe measures how far below, and with it the joules per TFLOP no kernel on this board can beat.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_power_probe.hip -o /tmp/mfma_power_probe -lpthread && /tmp/mfma_power_probe [waves per SIMD = 2] [seconds = 3] [0 random operands | 1 zeros | 2 operands read from LDS | 3 + LDS-DMA refills]
// The roofline fractions of DESIGN.md / bench.py stay against the nominal 2.5 PFLOP/s; this number only says how much of the gap is the power cap.
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include <dirent.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

// the same multiply with its operands READ FROM LDS the way the engine's 64 x 64 wave tiles read them: 8 fragment reads (ds_read_b128) per 16 MFMAs = 0.5 per MFMA
// (mode 1), and additionally refilled by LDS-DMA from a 16 MiB buffer that stays in L2 / Infinity Cache, 32 KiB per 128 MFMAs and workgroup (mode 2: a K-step of the
// implicit GEMM without its HBM traffic)
typedef __attribute__((ext_vector_type(4))) unsigned pp_u32x4;
__global__ __launch_bounds__(256) void mfma_lds_loop(float* __restrict__ sink, const char* __restrict__ src, int iters, int mode) {
    __shared__ __attribute__((aligned(16))) char lds[32768];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned s = (unsigned)(blockIdx.x * 256 + tid) * 40503u + 12345u;
    for (int i = tid; i < 32768 / 4; i += 256) {
        s = s * 1664525u + 1013904223u;
        const float f = (float)((s >> 9) & 0xffff) / 65536.f - 0.5f;
        ((unsigned*)lds)[i] = (__builtin_bit_cast(unsigned, f) & 0xffff0000u) | (__builtin_bit_cast(unsigned, f * 0.37f) >> 16);
    }
    __syncthreads();
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds + ((lane & 15) * 128 + ((lane >> 4) ^ (lane & 7)) * 16);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(src + (size_t)(blockIdx.x & 255) * 65536), 0, 65536, 0x00020000);
    f32x4_t acc[4][4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            pp_u32x4 a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                asm volatile("ds_read_b128 %0, %1" : "=v"(a[i]) : "v"(base + (unsigned)(i * 2048)) : "memory");
                asm volatile("ds_read_b128 %0, %1" : "=v"(b[i]) : "v"(base + (unsigned)(16384 + i * 2048)) : "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a[i]), __builtin_bit_cast(bf16x8_t, b[j]), acc[i][j], 0, 0, 0);
        }
        if (mode == 2 && (it & 1) == 1) {                  // every 128 MFMAs: 8 x 1 KiB per wave = 32 KiB per workgroup from L2 into LDS (contents stay random bf16)
#pragma unroll
            for (int q = 0; q < 8; ++q)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + (wave * 8 + q) * 1024), 16, (unsigned)(((it >> 1) & 1) * 32768 + (wave * 8 + q) * 1024 + lane * 16), 0, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }
    float t = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (t == 12345.678f) sink[lane] = t;
}

// 16 independent accumulator fragments per wave (64 registers): the MFMAs issue back to back
__global__ __launch_bounds__(256) void mfma_loop(float* __restrict__ sink, int iters, unsigned seed) {
    const int lane = threadIdx.x & 63;
    unsigned s = seed * 2654435761u + (unsigned)(blockIdx.x * 256 + threadIdx.x) * 40503u;
    bf16x8_t a[4], b[4];
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 8; ++e) {
            s = s * 1664525u + 1013904223u;
            a[i][e] = (__bf16)(seed ? ((float)((s >> 9) & 0xffff) / 65536.f - 0.5f) : 0.f);
            s = s * 1664525u + 1013904223u;
            b[i][e] = (__bf16)(seed ? ((float)((s >> 9) & 0xffff) / 65536.f - 0.5f) : 0.f);
        }
    f32x4_t acc[4][4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    float t = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (t == 12345.678f) sink[lane] = t;            // (keeps the loop alive)
}

static std::string hwmon_dir(int dev) {
    char bdf[64];
    if (hipDeviceGetPCIBusId(bdf, sizeof(bdf), dev) != hipSuccess) return "";
    for (char* c = bdf; *c; ++c) *c = (char)tolower(*c);
    const std::string base = std::string("/sys/bus/pci/devices/") + bdf + "/hwmon";
    DIR* d = opendir(base.c_str());
    if (!d) return "";
    std::string out;
    while (dirent* e = readdir(d))
        if (!strncmp(e->d_name, "hwmon", 5)) out = base + "/" + e->d_name;
    closedir(d);
    return out;
}
static double read_num(const std::string& path) {
    FILE* f = fopen(path.c_str(), "r");
    if (!f) return -1;
    double v = -1;
    if (fscanf(f, "%lf", &v) != 1) v = -1;
    fclose(f);
    return v;
}

int main(int argc, char** argv) {
    const int waves_per_simd = argc > 1 ? atoi(argv[1]) : 2;
    const double seconds = argc > 2 ? atof(argv[2]) : 3.0;
    const bool zeros = argc > 3 && atoi(argv[3]) == 1;
    const int lds_mode = argc > 3 && atoi(argv[3]) >= 2 ? atoi(argv[3]) - 1 : 0;       // 2: operands from LDS (0.5 reads per MFMA), 3: + LDS-DMA refills from L2
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const int grid = cus * waves_per_simd;            // 256 threads = 4 waves = one wave per SIMD per workgroup
    float* sink;
    hipMalloc(&sink, 4096);
    char* src = nullptr;
    hipMalloc(&src, 256 * 65536);
    {
        std::vector<unsigned> h(256 * 65536 / 4);
        unsigned s = 99u;
        for (auto& v : h) { s = s * 1664525u + 1013904223u; const float f = (float)((s >> 9) & 0xffff) / 65536.f - 0.5f; unsigned u; memcpy(&u, &f, 4); v = (u & 0xffff0000u) | (u >> 16); }
        hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    }
    const int iters = 20000;                          // 64 MFMAs per iteration and wave
    auto launch = [&]() {
        if (lds_mode) hipLaunchKernelGGL(mfma_lds_loop, dim3(grid), dim3(256), 0, 0, sink, src, iters, lds_mode);
        else hipLaunchKernelGGL(mfma_loop, dim3(grid), dim3(256), 0, 0, sink, iters, zeros ? 0u : 7u);
    };
    const double flop_per_launch = (double)grid * 4 * iters * 64 * (2.0 * 16 * 16 * 32);
    const std::string hw = hwmon_dir(0);
    std::atomic<bool> stop{false};
    std::vector<double> watts, mhz;
    std::thread sampler([&] {
        while (!stop.load()) {
            if (!hw.empty()) {
                const double p = read_num(hw + "/power1_input"), f = read_num(hw + "/freq1_input");
                if (p > 0) watts.push_back(p / 1e6);
                if (f > 0) mhz.push_back(f / 1e6);
            }
            std::this_thread::sleep_for(std::chrono::milliseconds(100));
        }
    });
    for (int i = 0; i < 3; ++i) launch();
    hipDeviceSynchronize();
    watts.clear(); mhz.clear();
    const auto t0 = std::chrono::steady_clock::now();
    long launches = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        for (int i = 0; i < 4; ++i) launch();
        launches += 4;
        hipDeviceSynchronize();
    }
    const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    stop.store(true);
    sampler.join();
    double w = 0, f = 0;
    for (double v : watts) w += v;
    for (double v : mhz) f += v;
    w = watts.empty() ? 0 : w / watts.size();
    f = mhz.empty() ? 0 : f / mhz.size();
    const double tflops = flop_per_launch * launches / el / 1e12;
    printf("| %s, %d wave(s) per SIMD | %.0f TFLOP/s | %.3f of 2500 | %.0f W | %.0f MHz | %.3f J/TFLOP | %.0f TFLOP/s at that clock's peak (%d CUs x 4096 FLOP/clk) |\n", lds_mode == 2 ? "random operands read from LDS (0.5 reads per MFMA) + LDS-DMA refills from L2" : (lds_mode == 1 ? "random operands read from LDS (0.5 reads per MFMA)" : (zeros ? "zero operands" : "random operands")),
           waves_per_simd, tflops, tflops / 2500.0, w, f, w > 0 ? w / tflops : 0.0, cus * 4096.0 * f * 1e6 / 1e12, cus);
    hipFree(sink);
    return 0;
}
