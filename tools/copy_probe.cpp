// copy_probe.cpp -- what a host<->device copy of one chunk of a component-major [rows][B] array costs on this box:
// hipMemcpy2DAsync (one call, pitch B) against `rows` 1-D copies against one contiguous copy of the same bytes; pinned host memory.
//   hipcc -O2 tools/copy_probe.cpp -o tools/copy_probe && tools/copy_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t B = 65536, rows = 28, chunk = 8192;
    double *h = nullptr, *d = nullptr;
    CK(hipHostMalloc(reinterpret_cast<void **>(&h), rows * B * 8, hipHostMallocDefault));
    CK(hipMalloc(reinterpret_cast<void **>(&d), rows * B * 8));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (int dir = 0; dir < 2; ++dir) {
        const hipMemcpyKind kind = dir == 0 ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost;
        for (int rep = 0; rep < 3; ++rep) {
            double t = now();
            for (size_t k = 0; k < B / chunk; ++k) {
                void *dst = dir == 0 ? static_cast<void *>(d + k * chunk * rows) : static_cast<void *>(h + k * chunk);
                const void *src = dir == 0 ? static_cast<const void *>(h + k * chunk) : static_cast<const void *>(d + k * chunk * rows);
                CK(hipMemcpy2DAsync(dst, dir == 0 ? chunk * 8 : B * 8, src, dir == 0 ? B * 8 : chunk * 8, chunk * 8, rows, kind, s));
            }
            double ti = now() - t;
            CK(hipStreamSynchronize(s));
            printf("%s 2D  x%zu chunks: issue %.3f ms, done %.3f ms\n", dir == 0 ? "H2D" : "D2H", B / chunk, ti * 1e3, (now() - t) * 1e3);
            t = now();
            for (size_t k = 0; k < B / chunk; ++k)
                for (size_t r = 0; r < rows; ++r) {
                    void *dst = dir == 0 ? static_cast<void *>(d + (k * rows + r) * chunk) : static_cast<void *>(h + r * B + k * chunk);
                    const void *src = dir == 0 ? static_cast<const void *>(h + r * B + k * chunk) : static_cast<const void *>(d + (k * rows + r) * chunk);
                    CK(hipMemcpyAsync(dst, src, chunk * 8, kind, s));
                }
            ti = now() - t;
            CK(hipStreamSynchronize(s));
            printf("%s 1D rows x%zu: issue %.3f ms, done %.3f ms\n", dir == 0 ? "H2D" : "D2H", B / chunk * rows, ti * 1e3, (now() - t) * 1e3);
            t = now();
            CK(hipMemcpyAsync(dir == 0 ? static_cast<void *>(d) : static_cast<void *>(h), dir == 0 ? static_cast<const void *>(h) : static_cast<const void *>(d), rows * B * 8, kind, s));
            ti = now() - t;
            CK(hipStreamSynchronize(s));
            printf("%s one contiguous copy: issue %.3f ms, done %.3f ms (%.1f GB/s)\n", dir == 0 ? "H2D" : "D2H", ti * 1e3, (now() - t) * 1e3, rows * B * 8 / (now() - t) / 1e9);
        }
    }
    return 0;
}
