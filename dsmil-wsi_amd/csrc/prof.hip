#include "prof.h"
#include <stdint.h>
#include "dsmil_hip.h"

namespace dsmil_prof {
namespace {
constexpr int RING = 4096;
struct Chan {
    int n = 0;
    hipEvent_t e0[RING], e1[RING];
};
bool g_on = false, g_created = false;
Chan g_ch[NCH];
}  // namespace

int begin(int channel, hipStream_t st) {
    if (!g_on) return -1;
    Chan& c = g_ch[channel];
    if (c.n >= RING) return -1;
    (void)hipEventRecord(c.e0[c.n], st);
    return c.n++;
}
void end(int channel, int slot, hipStream_t st) {
    if (slot >= 0) (void)hipEventRecord(g_ch[channel].e1[slot], st);
}
}  // namespace dsmil_prof

extern "C" {
int dsmil_profile_enable(int on) {
    using namespace dsmil_prof;
    if (on && !g_created) {
        for (int c = 0; c < NCH; ++c)
            for (int i = 0; i < RING; ++i) {
                if (hipEventCreate(&g_ch[c].e0[i]) != hipSuccess) return DSMIL_E_LAUNCH;
                if (hipEventCreate(&g_ch[c].e1[i]) != hipSuccess) return DSMIL_E_LAUNCH;
            }
        g_created = true;
    }
    g_on = on != 0;
    for (int c = 0; c < NCH; ++c) g_ch[c].n = 0;
    return DSMIL_OK;
}

int dsmil_profile_collect(int channel, double* total_ms, int64_t* launches) {
    using namespace dsmil_prof;
    if (!total_ms || !launches || channel < 0 || channel >= NCH) return DSMIL_E_INVALID;
    Chan& c = g_ch[channel];
    double t = 0.0;
    for (int i = 0; i < c.n; ++i) {
        float ms = 0.f;
        if (hipEventSynchronize(c.e1[i]) != hipSuccess) return DSMIL_E_LAUNCH;
        if (hipEventElapsedTime(&ms, c.e0[i], c.e1[i]) != hipSuccess) return DSMIL_E_LAUNCH;
        t += ms;
    }
    *total_ms = t;
    *launches = c.n;
    c.n = 0;
    return DSMIL_OK;
}
}
