// xfer.hip -- host<->device copy rates of the box, by method, to size the host-pointer entry
// points' staging (host_api.hip).  hipcc --offload-arch=gfx950 -O2 -o xfer xfer.hip -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void pcopy(uint8_t *d, const uint8_t *s, size_t n, int threads) {
    if (threads <= 1) { memcpy(d, s, n); return; }
    std::vector<std::thread> t;
    size_t per = (n / threads + 4095) & ~(size_t)4095;
    for (int i = 0; i < threads; i++) {
        size_t a = (size_t)i * per; if (a >= n) break;
        size_t m = n - a < per ? n - a : per;
        t.emplace_back([=] { memcpy(d + a, s + a, m); });
    }
    for (auto &x : t) x.join();
}
int main(int argc, char **argv) {
    size_t n = (size_t)(argc > 1 ? atof(argv[1]) : 1.0) * (1u << 30);
    uint8_t *h = (uint8_t *)malloc(n), *h2 = (uint8_t *)malloc(n), *d;
    memset(h, 1, n); memset(h2, 2, n);
    CK(hipMalloc(&d, n));
    CK(hipMemcpy(d, h, n, hipMemcpyHostToDevice));
    for (int rep = 0; rep < 2; rep++) {
        double t = now(); CK(hipMemcpy(d, h, n, hipMemcpyHostToDevice)); printf("pageable H2D hipMemcpy      %6.2f GB/s\n", n / (now() - t) / 1e9);
        t = now(); CK(hipMemcpy(h2, d, n, hipMemcpyDeviceToHost)); printf("pageable D2H hipMemcpy      %6.2f GB/s\n", n / (now() - t) / 1e9);
    }
    { double t = now(); hipError_t e = hipHostRegister(h, n, hipHostRegisterDefault); double tr = now() - t;
      if (e != hipSuccess) { printf("hipHostRegister of malloc'd memory: %s\n", hipGetErrorString(e)); (void)hipGetLastError(); }
      else { t = now(); CK(hipMemcpy(d, h, n, hipMemcpyHostToDevice)); double tc = now() - t; CK(hipHostUnregister(h));
             printf("hipHostRegister %.1f ms, registered H2D %.2f GB/s\n", tr * 1e3, n / tc / 1e9); } }
    for (size_t piece : {(size_t)8 << 20, (size_t)32 << 20}) for (int threads : {1, 2, 4, 8}) {
        uint8_t *p[2]; hipEvent_t ev[2]; hipStream_t st; CK(hipStreamCreate(&st));
        for (int i = 0; i < 2; i++) { CK(hipHostMalloc(&p[i], piece)); memset(p[i], 0, piece); CK(hipEventCreate(&ev[i])); }
        double t = now(); int k = 0;
        for (size_t a = 0; a < n; a += piece, k ^= 1) {
            size_t m = n - a < piece ? n - a : piece;
            CK(hipEventSynchronize(ev[k]));
            pcopy(p[k], h + a, m, threads);
            CK(hipMemcpyAsync(d + a, p[k], m, hipMemcpyHostToDevice, st));
            CK(hipEventRecord(ev[k], st));
        }
        CK(hipStreamSynchronize(st));
        printf("staged H2D piece %2zu MiB threads %d  %6.2f GB/s\n", piece >> 20, threads, n / (now() - t) / 1e9);
        t = now(); k = 0; size_t pa[2] = {0, 0}, pm[2] = {0, 0};
        for (size_t a = 0; a < n + piece; a += piece, k ^= 1) {
            // drain the piece issued two iterations ago from this buffer, then reuse it
            if (pm[k]) { CK(hipEventSynchronize(ev[k])); pcopy(h2 + pa[k], p[k], pm[k], threads); pm[k] = 0; }
            if (a < n) { size_t m = n - a < piece ? n - a : piece; CK(hipMemcpyAsync(p[k], d + a, m, hipMemcpyDeviceToHost, st)); CK(hipEventRecord(ev[k], st)); pa[k] = a; pm[k] = m; }
        }
        k ^= 1; if (pm[k]) { CK(hipEventSynchronize(ev[k])); pcopy(h2 + pa[k], p[k], pm[k], threads); }
        printf("staged D2H piece %2zu MiB threads %d  %6.2f GB/s  %s\n", piece >> 20, threads, n / (now() - t) / 1e9, memcmp(h, h2, n) ? "MISMATCH" : "ok");
        for (int i = 0; i < 2; i++) { CK(hipHostFree(p[i])); CK(hipEventDestroy(ev[i])); } CK(hipStreamDestroy(st));
    }
    { uint8_t *p; CK(hipHostMalloc(&p, n)); memset(p, 3, n);
      double t = now(); CK(hipMemcpy(d, p, n, hipMemcpyHostToDevice)); printf("pinned H2D                  %6.2f GB/s\n", n / (now() - t) / 1e9);
      t = now(); CK(hipMemcpy(p, d, n, hipMemcpyDeviceToHost)); printf("pinned D2H                  %6.2f GB/s\n", n / (now() - t) / 1e9);
      double t0 = now(); pcopy(h2, p, n, 8); printf("host memcpy 8 threads       %6.2f GB/s, 1 thread ", n / (now() - t0) / 1e9); t0 = now(); memcpy(h2, p, n); printf("%6.2f GB/s\n", n / (now() - t0) / 1e9); }
    return 0;
}
