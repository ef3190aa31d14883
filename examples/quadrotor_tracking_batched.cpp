// quadrotor_tracking_batched.cpp — the MPC loop of the reference's examples/quadrotor_tracking.cpp (:93-118), for B
// quadrotors at once, written against the C-ABI only (include/tinympc_batch.h).  Plain C++17, no HIP/Eigen/torch types:
//
//   g++ -std=c++17 -O2 -Iinclude examples/quadrotor_tracking_batched.cpp -Laccelerated-tinympc_amd/lib -ltinympc_hip
//       -Wl,-rpath,$PWD/accelerated-tinympc_amd/lib -o build/quadrotor_tracking_batched
//   ./build/quadrotor_tracking_batched accelerated-tinympc_amd/data/quadrotor_20hz.bin 4096 100
//
// The problem data file is the flat binary written by `python tools/export_problem_bin.py` (row-major doubles of
// rho, Adyn, Bdyn, Kinf, Pinf, Quu_inv, AmBKt, Q — the numbers of examples/problem_data/quadrotor_20hz_params.hpp).
// Every instance tracks the y_axis_line trajectory from its own start index; the loop stays on the device
// (tiny_batch_mpc_step_async), the host only reads back the states to print the tracking error like the reference does.
#include "tinympc_batch.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

static constexpr int NX = 12, NU = 4, N = 30, NTOTAL = 301;

#define CHECK(call)                                                                        \
    do                                                                                     \
    {                                                                                      \
        int rc_ = (call);                                                                  \
        if (rc_ < 0)                                                                       \
        {                                                                                  \
            std::fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, tiny_batch_last_error()); \
            return 1;                                                                      \
        }                                                                                  \
    } while (0)

// row-major (as in the reference's headers) -> column-major float (Eigen's storage, what the ABI takes)
static std::vector<float> colmajor(const double *rm, int rows, int cols)
{
    std::vector<float> cm((size_t)rows * cols);
    for (int i = 0; i < rows; i++)
        for (int j = 0; j < cols; j++) cm[(size_t)j * rows + i] = (float)rm[(size_t)i * cols + j];
    return cm;
}

int main(int argc, char **argv)
{
    if (argc < 2) { std::fprintf(stderr, "usage: %s quadrotor_20hz.bin [batch] [steps]\n", argv[0]); return 2; }
    const int B = argc > 2 ? std::atoi(argv[2]) : 1024, steps = argc > 3 ? std::atoi(argv[3]) : 50;
    const size_t ndbl = 1 + NX * NX + NX * NU + NU * NX + NX * NX + NU * NU + NX * NX + NX;
    std::vector<double> raw(ndbl);
    FILE *f = std::fopen(argv[1], "rb");
    if (!f || std::fread(raw.data(), sizeof(double), ndbl, f) != ndbl) { std::fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
    std::fclose(f);
    const double *p = raw.data();
    const float rho = (float)*p++;
    const auto A = colmajor(p, NX, NX); p += NX * NX;
    const auto Bd = colmajor(p, NX, NU); p += NX * NU;
    const auto K = colmajor(p, NU, NX); p += NU * NX;
    const auto Pinf = colmajor(p, NX, NX); p += NX * NX;
    const auto Qi = colmajor(p, NU, NU); p += NU * NU;
    const auto Am = colmajor(p, NX, NX); p += NX * NX;
    std::vector<float> Q(p, p + NX);

    // reference trajectory: z = 1 m, y from 0 to 4 m, dy = 0.2666667 m/s (quadrotor_20hz_y_axis_line.hpp)
    std::vector<float> table((size_t)NTOTAL * NX, 0.f);
    for (int k = 0; k < NTOTAL; k++)
    {
        table[(size_t)k * NX + 1] = (float)(std::round(k * 4.0 / 300.0 * 1e7) / 1e7);
        table[(size_t)k * NX + 2] = 1.f;
        table[(size_t)k * NX + 7] = k < NTOTAL - 1 ? 0.2666667f : 0.f;
    }
    std::vector<int> start(B);
    std::vector<float> x0((size_t)B * NX);
    for (int b = 0; b < B; b++)
    {
        start[b] = b % (NTOTAL - N - steps > 0 ? NTOTAL - N - steps : 1);
        for (int i = 0; i < NX; i++) x0[(size_t)b * NX + i] = table[(size_t)start[b] * NX + i]; // x0 = Xref.col(0) (tracking.cpp:88)
    }
    std::vector<float> xmin((size_t)N * NX, -5.f), xmax((size_t)N * NX, 5.f), umin((size_t)(N - 1) * NU, -0.5f), umax((size_t)(N - 1) * NU, 0.5f);

    TinyBatch *tb = nullptr;
    CHECK(tiny_batch_create(&tb, NX, NU, N, B, 0));
    CHECK(tiny_batch_set_cache(tb, rho, K.data(), Pinf.data(), Qi.data(), Am.data()));
    CHECK(tiny_batch_set_dynamics(tb, A.data(), Bd.data(), Q.data()));
    CHECK(tiny_batch_set_settings(tb, 1e-3f, 1e-3f, 100, 1, 1, 1)); // quadrotor_tracking.cpp:75-80
    CHECK(tiny_batch_set_xmin(tb, xmin.data(), 1)); CHECK(tiny_batch_set_xmax(tb, xmax.data(), 1));
    CHECK(tiny_batch_set_umin(tb, umin.data(), 1)); CHECK(tiny_batch_set_umax(tb, umax.data(), 1));
    CHECK(tiny_batch_set_xref_window(tb, table.data(), NTOTAL, start.data()));
    CHECK(tiny_batch_set_x0(tb, x0.data()));
    std::printf("kernel: %s, %d instances\n", tiny_batch_kernel_name(tb), B);

    std::vector<float> x((size_t)B * NX);
    std::vector<int> iters(B);
    for (int k = 0; k < steps; k++)
    {
        CHECK(tiny_batch_mpc_step_async(tb, /*window_advance=*/1)); // x0 -> solve -> x0 = A x0 + B u0, window slides by one
        CHECK(tiny_batch_get_x0(tb, x.data()));
        CHECK(tiny_batch_get_status(tb, iters.data(), nullptr, nullptr));
        double err = 0, it = 0;
        for (int b = 0; b < B; b++)
        {
            double e2 = 0;
            for (int i = 0; i < NX; i++)
            {
                const double dlt = x[(size_t)b * NX + i] - table[(size_t)(start[b] + k + 1) * NX + i];
                e2 += dlt * dlt;
            }
            err += std::sqrt(e2);
            it += iters[b];
        }
        if (k < 5 || k % 10 == 9) std::printf("step %3d: mean tracking error %.6f, mean ADMM iterations %.2f\n", k, err / B, it / B);
    }
    tiny_batch_destroy(tb);
    return 0;
}
