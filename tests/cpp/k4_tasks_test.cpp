// Host-only test of the task list of the Schur kernel (k4_build_tasks, cvx_amd/csrc/calib_kernels.h): for every width
// the kernel takes, the tasks must cover each 16x16 block of the upper triangle exactly once, and in every workgroup
// the sixteen rows of a stage must be copied into LDS exactly once, never by a DIAG wave.  No device work.
// Built by cvx_amd/csrc/Makefile (hipcc: the header holds the kernels too), run by tests/test_capi_and_host.py.
#include <cstdio>
#include <vector>

#include "../../cvx_amd/csrc/calib_kernels.h"

using namespace cvxk;

int main() {
    int bad = 0;
    for (int nb = 1; nb <= K4_PITCH / 16; ++nb) {
        std::vector<K4Task> tasks;
        const int parts = k4_build_tasks(nb, tasks);
        if (parts <= 0 || (int)tasks.size() != parts * K4_WAVES) { printf("{\"nb\": %d, \"error\": \"no task list\"}\n", nb); ++bad; continue; }
        std::vector<int> cover((size_t)nb * nb, 0);
        auto hit = [&](int gA, int gB) { if (gA >= 0 && gB >= gA && gB < nb) cover[(size_t)gA * nb + gB]++; else ++bad; };
        for (const K4Task &t : tasks) {
            if (t.nA == 0) continue;
            if (!t.diag) {
                for (int i = 0; i < t.nA; ++i) for (int j = 0; j < t.nB; ++j) hit(t.a0 + i, t.b0 + j);
            } else {
                for (int i = 0; i < t.nA; ++i) for (int j = i; j < t.nB; ++j) hit(t.a0 + i, t.b0 + j);
                if (t.edge >= 0) {
                    for (int i = 0; i < t.nA; ++i) hit(t.a0 + i, t.edge);
                    if (t.corner) hit(t.edge, t.edge);
                }
            }
        }
        int wrong = 0;
        for (int i = 0; i < nb; ++i) for (int j = i; j < nb; ++j) wrong += cover[(size_t)i * nb + j] != 1;
        int copy_wrong = 0, diag_copies = 0;
        for (int p = 0; p < parts; ++p) {
            int rows[K4_KB] = {0};
            for (int w = 0; w < K4_WAVES; ++w) {
                const K4Task &t = tasks[(size_t)p * K4_WAVES + w];
                const int first = t.copy & 255, cnt = t.copy >> 8;
                if (cnt > 0 && t.nA > 0 && t.diag) ++diag_copies;
                for (int r = first; r < first + cnt; ++r) { if (r < K4_KB) rows[r]++; else ++copy_wrong; }
            }
            for (int r = 0; r < K4_KB; ++r) copy_wrong += rows[r] != 1;
        }
        printf("{\"nb\": %d, \"parts\": %d, \"blocks_wrong\": %d, \"copy_wrong\": %d, \"diag_copies\": %d}\n", nb, parts, wrong,
               copy_wrong, diag_copies);
        bad += wrong + copy_wrong + diag_copies;
    }
    return bad ? 1 : 0;
}
