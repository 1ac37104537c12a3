// host-side staging rates: contiguous and column-slab (strided) copies with T threads
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv)
{
    const size_t rows = 1 << 19, row = 4096;
    char* a = (char*)malloc(rows * row);
    char* b = (char*)malloc(rows * row);
    char* c = (char*)malloc(rows * row);
    char* d = (char*)malloc(rows * row);
    memset(a, 1, rows * row); memset(b, 2, rows * row); memset(c, 3, rows * row); memset(d, 4, rows * row);
    for (int T : {2, 4, 6, 8}) {
        for (size_t piece : {4096, 2048, 1024, 512}) {
            // gather: slab h of every row of a -> compact region of b; at the same time scatter compact c -> slab of d (other T threads)
            const size_t H = row / piece;
            double t0 = now();
            std::vector<std::thread> th;
            for (int t = 0; t < T; t++)
                th.emplace_back([=] {
                    for (size_t h = 0; h < H; h++) {
                        char* dst = b + h * rows * piece;
                        for (size_t r = t; r < rows; r += T) memcpy(dst + r * piece, a + r * row + h * piece, piece);
                    }
                });
            for (int t = 0; t < T; t++)
                th.emplace_back([=] {
                    for (size_t h = 0; h < H; h++) {
                        const char* src = c + h * rows * piece;
                        for (size_t r = t; r < rows; r += T) memcpy(d + r * row + h * piece, src + r * piece, piece);
                    }
                });
            for (auto& x : th) x.join();
            double dt = now() - t0;
            printf("T=%d+%d piece=%zu: 2 GiB gathered + 2 GiB scattered in %.1f ms (%.1f GB/s each way)\n", T, T, piece, dt * 1e3, rows * row / dt / 1e9);
            fflush(stdout);
        }
    }
    return 0;
}
