// quadrotor_tracking_multigpu.cpp — the batch of examples/quadrotor_tracking_batched.cpp sharded over the GPUs of one node from
// ONE host thread, through the C-ABI only (include/tinympc_batch.h): plain C++17, no HIP / Eigen / torch types.
//
//   g++ -std=c++17 -O2 -Iinclude examples/quadrotor_tracking_multigpu.cpp -Laccelerated-tinympc_amd/lib -ltinympc_hip
//       -Wl,-rpath,$PWD/accelerated-tinympc_amd/lib -o build/quadrotor_tracking_multigpu
//   ./build/quadrotor_tracking_multigpu accelerated-tinympc_amd/data/quadrotor_20hz.bin 65536 0,1,2,3,4,5,6,7
//   ./build/quadrotor_tracking_multigpu accelerated-tinympc_amd/data/quadrotor_20hz.bin 4096 0,0      (two handles on one GPU)
//
// instances are independent (src/tinympc/admm.cpp touches one workspace), so handle g owns the contiguous
// block [g*B/G, (g+1)*B/G) of the instance index on device g with the gains and settings replicated; no data-path collective.
// tiny_batch_group_solve launches every device's solve before waiting for any (one stream per handle);
// tiny_batch_group_get_u0 gathers u.col(0) of all blocks device-to-device (hipMemcpyPeerAsync, xGMI) into one buffer.
// The program then solves the whole batch with ONE handle on the first device and checks that the sharded results are
// bit-identical (u.col(0), iteration counts, status): exit code 0 only then.
#include "tinympc_batch.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
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

static std::vector<float> colmajor(const double *rm, int rows, int cols)
{
    std::vector<float> cm((size_t)rows * cols);
    for (int i = 0; i < rows; i++)
        for (int j = 0; j < cols; j++) cm[(size_t)j * rows + i] = (float)rm[(size_t)i * cols + j];
    return cm;
}

struct Problem
{
    float rho;
    std::vector<float> A, Bd, K, Pinf, Qi, Am, Q, table, xmin, xmax, umin, umax;
};

// one handle for the instances [lo, hi) of the global batch on `device`
static int make_handle(TinyBatch **out, const Problem &P, int device, int lo, int hi, const std::vector<float> &x0, const std::vector<int> &start)
{
    TinyBatch *tb = nullptr;
    CHECK(tiny_batch_create(&tb, NX, NU, N, hi - lo, device));
    CHECK(tiny_batch_set_cache(tb, P.rho, P.K.data(), P.Pinf.data(), P.Qi.data(), P.Am.data()));
    CHECK(tiny_batch_set_dynamics(tb, P.A.data(), P.Bd.data(), P.Q.data()));
    CHECK(tiny_batch_set_settings(tb, 1e-3f, 1e-3f, 100, 1, 1, 1)); // quadrotor_tracking.cpp:75-80
    CHECK(tiny_batch_set_xmin(tb, P.xmin.data(), 1)); CHECK(tiny_batch_set_xmax(tb, P.xmax.data(), 1));
    CHECK(tiny_batch_set_umin(tb, P.umin.data(), 1)); CHECK(tiny_batch_set_umax(tb, P.umax.data(), 1));
    CHECK(tiny_batch_set_xref_window(tb, P.table.data(), NTOTAL, start.data() + lo));
    CHECK(tiny_batch_set_x0(tb, x0.data() + (size_t)lo * NX));
    *out = tb;
    return 0;
}

int main(int argc, char **argv)
{
    if (argc < 2) { std::fprintf(stderr, "usage: %s quadrotor_20hz.bin [total batch] [device list, e.g. 0,1,2,3]\n", argv[0]); return 2; }
    const int B = argc > 2 ? std::atoi(argv[2]) : 4096;
    std::vector<int> devices;
    {
        std::string list = argc > 3 ? argv[3] : "0";
        size_t pos = 0;
        while (pos <= list.size())
        {
            const size_t c = list.find(',', pos);
            devices.push_back(std::atoi(list.substr(pos, c == std::string::npos ? std::string::npos : c - pos).c_str()));
            if (c == std::string::npos) break;
            pos = c + 1;
        }
    }
    const int G = (int)devices.size();
    if (B < G) { std::fprintf(stderr, "batch %d smaller than the number of handles %d\n", B, G); return 2; }
    const size_t ndbl = 1 + NX * NX + NX * NU + NU * NX + NX * NX + NU * NU + NX * NX + NX;
    std::vector<double> raw(ndbl);
    FILE *f = std::fopen(argv[1], "rb");
    if (!f || std::fread(raw.data(), sizeof(double), ndbl, f) != ndbl) { std::fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
    std::fclose(f);
    Problem P;
    const double *p = raw.data();
    P.rho = (float)*p++;
    P.A = colmajor(p, NX, NX); p += NX * NX;
    P.Bd = colmajor(p, NX, NU); p += NX * NU;
    P.K = colmajor(p, NU, NX); p += NU * NX;
    P.Pinf = colmajor(p, NX, NX); p += NX * NX;
    P.Qi = colmajor(p, NU, NU); p += NU * NU;
    P.Am = colmajor(p, NX, NX); p += NX * NX;
    P.Q.assign(p, p + NX);
    P.table.assign((size_t)NTOTAL * NX, 0.f); // y_axis_line: z = 1 m, y from 0 to 4 m (quadrotor_20hz_y_axis_line.hpp)
    for (int k = 0; k < NTOTAL; k++)
    {
        P.table[(size_t)k * NX + 1] = (float)(std::round(k * 4.0 / 300.0 * 1e7) / 1e7);
        P.table[(size_t)k * NX + 2] = 1.f;
        P.table[(size_t)k * NX + 7] = k < NTOTAL - 1 ? 0.2666667f : 0.f;
    }
    P.xmin.assign((size_t)N * NX, -5.f); P.xmax.assign((size_t)N * NX, 5.f);
    P.umin.assign((size_t)(N - 1) * NU, -0.5f); P.umax.assign((size_t)(N - 1) * NU, 0.5f);
    std::vector<int> start(B);
    std::vector<float> x0((size_t)B * NX);
    unsigned lcg = 20241024u; // a fixed perturbation of x0 = Xref.col(0), so that the instances differ
    for (int b = 0; b < B; b++)
    {
        start[b] = b % (NTOTAL - N);
        for (int i = 0; i < NX; i++)
        {
            lcg = lcg * 1664525u + 1013904223u;
            x0[(size_t)b * NX + i] = P.table[(size_t)start[b] * NX + i] + ((lcg >> 8) * (1.0f / 16777216.0f) - 0.5f) * 0.1f;
        }
    }

    // ---- sharded: one handle per listed device, contiguous blocks ----
    std::vector<TinyBatch *> hs(G, nullptr);
    std::vector<int> lo(G), hi(G);
    for (int g = 0; g < G; g++)
    {
        lo[g] = (int)((long long)B * g / G);
        hi[g] = (int)((long long)B * (g + 1) / G);
        if (make_handle(&hs[g], P, devices[g], lo[g], hi[g], x0, start)) return 1;
        std::printf("handle %d: device %d, instances [%d, %d), kernel %s\n", g, devices[g], lo[g], hi[g], tiny_batch_kernel_name(hs[g]));
    }
    int unsolved = 0;
    CHECK(tiny_batch_group_solve(hs.data(), G, &unsolved));
    std::vector<float> u0((size_t)B * NU);
    CHECK(tiny_batch_group_get_u0(hs.data(), G, u0.data()));
    std::vector<int> iters(B), status(B);
    for (int g = 0; g < G; g++) CHECK(tiny_batch_get_status(hs[g], iters.data() + lo[g], status.data() + lo[g], nullptr));
    for (int g = 0; g < G; g++) tiny_batch_destroy(hs[g]);

    // ---- the same batch through ONE handle on the first device ----
    TinyBatch *one = nullptr;
    if (make_handle(&one, P, devices[0], 0, B, x0, start)) return 1;
    CHECK(tiny_batch_solve(one));
    std::vector<float> u_all((size_t)B * (N - 1) * NU);
    CHECK(tiny_batch_get_u(one, u_all.data()));
    std::vector<int> iters1(B), status1(B);
    CHECK(tiny_batch_get_status(one, iters1.data(), status1.data(), nullptr));
    tiny_batch_destroy(one);

    long long bad = 0, itsum = 0;
    for (int b = 0; b < B; b++)
    {
        itsum += iters[b];
        if (iters[b] != iters1[b] || status[b] != status1[b]) bad++;
        else if (std::memcmp(&u0[(size_t)b * NU], &u_all[(size_t)b * (N - 1) * NU], NU * sizeof(float)) != 0) bad++;
    }
    std::printf("%d instances over %d handles: %d hit max_iter, mean %.2f ADMM iterations; u.col(0), iter and status of %lld instances differ from the single-handle solve\n",
                B, G, unsolved, (double)itsum / B, bad);
    std::printf(bad == 0 ? "sharded == single handle, bit for bit\n" : "MISMATCH\n");
    return bad == 0 ? 0 : 1;
}
