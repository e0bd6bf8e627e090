// hipMalloc / hipFree cost: one block of N MiB against k blocks of N / k MiB (context creation)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    hipFree(nullptr);
    for (int rep = 0; rep < 2; rep++)
        for (size_t mib : {64, 1024, 4096}) {
            for (int k : {1, 16, 96}) {
                std::vector<void*> p(k);
                double t0 = now();
                for (int i = 0; i < k; i++)
                    if (hipMalloc(&p[i], (mib << 20) / k) != hipSuccess) { std::printf("alloc failed\n"); return 1; }
                double t1 = now();
                for (int i = 0; i < k; i++) (void)hipFree(p[i]);
                double t2 = now();
                std::printf("%5zu MiB in %2d blocks: malloc %7.3f ms, free %7.3f ms\n", mib, k, t1 - t0, t2 - t1);
            }
        }
    return 0;
}
