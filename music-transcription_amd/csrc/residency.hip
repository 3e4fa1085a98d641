// Admission control for the persistent (self-synchronising) launches of libmt_hip.so: the forward recurrences (lstm.hip)
// and the backward recurrence (lstm_bwd.hip) wait on their OWN workgroups, so every workgroup of every such launch in flight
// on a GPU must be resident at the same time.  Launches on ONE stream run one after the other; launches on different
// streams may overlap.  The library therefore keeps, per device, the streams that have a persistent launch pending (an
// event recorded behind each launch tells when it has drained) with the number of CUs that launch needs (workgroups divided
// by the occupancy query's workgroups per CU), and refuses -- MT_EUNSUPPORTED, immediately, instead of a 2-second spin
// timeout per layer later -- a launch whose CUs, added to those of the other streams' pending launches, exceed the device.
// This is the only mutable global state of the library (a mutex-protected table of streams and events).
#include "mt_common.h"
#include <mutex>
#include <vector>

namespace mt {

struct PersistentEntry {
    hipStream_t st;
    hipEvent_t ev;
    double need;      // CUs
    bool pending;     // a launch is queued on the stream, `ev` is recorded behind it
    int reserved;     // launches admitted but not yet committed by persistent_mark (or given up by persistent_cancel)
    double rneed;     // CUs of the reservation(s)
};
static std::mutex g_mu;
static std::vector<PersistentEntry> g_tab[16];
static int g_ncu[16];

int persistent_admit(const void* kernel, int block, size_t smem, int nwg, hipStream_t st, const char* who) {
    int dev = 0;
    MT_CHECK_HIP(hipGetDevice(&dev));
    MT_REQUIRE(dev >= 0 && dev < 16, MT_EUNSUPPORTED, "%s: device index %d", who, dev);
    int per_cu = 0;
    MT_CHECK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, block, smem));
    MT_REQUIRE(per_cu >= 1, MT_EUNSUPPORTED, "%s: the kernel does not fit a CU", who);
    std::lock_guard<std::mutex> lock(g_mu);
    if (g_ncu[dev] == 0) MT_CHECK_HIP(hipDeviceGetAttribute(&g_ncu[dev], hipDeviceAttributeMultiprocessorCount, dev));
    const double need = (double)nwg / per_cu, cap = (double)g_ncu[dev];
    MT_REQUIRE(need <= cap, MT_EUNSUPPORTED, "%s: %d workgroups at %d per CU cannot be resident on %d CUs", who, nwg, per_cu, g_ncu[dev]);
    double others = 0.0;
    PersistentEntry* mine = nullptr;
    // Admission is check-AND-RESERVE under one lock: the CUs of an admitted launch count against other host threads' admissions
    // from here on, whether or not its event has been recorded yet (persistent_mark commits the reservation, persistent_cancel
    // gives it up when the launch failed).  Entries of streams whose work has drained hold no CUs and are reused when the
    // runtime hands the same stream handle out again; nothing else outlives a launch.
    for (auto& e : g_tab[dev]) {
        if (e.pending && hipEventQuery(e.ev) == hipSuccess) { e.pending = false; e.need = 0.0; }
        if (e.st == st) mine = &e;
        else others += (e.pending ? e.need : 0.0) > e.rneed ? (e.pending ? e.need : 0.0) : e.rneed;
    }
    (void)hipGetLastError();          // hipEventQuery reports "not ready" as an error code: clear it
    if (others + need > cap + 1e-9) {
        set_error("%s: %.0f CUs for this persistent launch + %.0f CUs held by persistent launches pending on other streams > %d CUs: "
                  "they could not all be resident and would stall on each other (fewer forwards in flight per GPU)",
                  who, need, others, g_ncu[dev]);
        return MT_EUNSUPPORTED;
    }
    if (!mine) {
        PersistentEntry e{st, nullptr, 0.0, false, 0, 0.0};
        MT_CHECK_HIP(hipEventCreateWithFlags(&e.ev, hipEventDisableTiming));
        g_tab[dev].push_back(e);
        mine = &g_tab[dev].back();
    }
    mine->reserved += 1;
    mine->rneed = mine->rneed > need ? mine->rneed : need;      // launches on one stream run one after the other: the largest counts
    return MT_OK;
}

static PersistentEntry* find_entry(int dev, hipStream_t st) {
    for (auto& e : g_tab[dev])
        if (e.st == st) return &e;
    return nullptr;
}

// the admitted launch failed: give the reservation back
int persistent_cancel(hipStream_t st) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return MT_EHIP;
    std::lock_guard<std::mutex> lock(g_mu);
    if (PersistentEntry* e = find_entry(dev, st)) {
        if (e->reserved > 0 && --e->reserved == 0) e->rneed = 0.0;
    }
    return MT_OK;
}

// after the launch: the stream's entry stays pending until everything queued so far on the stream has drained
int persistent_mark(hipStream_t st) {
    int dev = 0;
    MT_CHECK_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_mu);
    if (PersistentEntry* e = find_entry(dev, st)) {
        const double was = e->pending ? e->need : 0.0;
        MT_CHECK_HIP(hipEventRecord(e->ev, st));
        e->need = was > e->rneed ? was : e->rneed;
        e->pending = true;
        if (e->reserved > 0 && --e->reserved == 0) e->rneed = 0.0;
    }
    return MT_OK;
}

}  // namespace mt

// CUs held by persistent launches pending on streams other than `stream` (diagnostics / tests); negative on error.
extern "C" int mt_persistent_cus_in_flight(mt_stream_t stream) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return MT_EHIP;
    std::lock_guard<std::mutex> lock(mt::g_mu);
    double others = 0.0;
    for (auto& e : mt::g_tab[dev]) {
        if (e.pending && hipEventQuery(e.ev) == hipSuccess) { e.pending = false; e.need = 0.0; }
        if (e.st != (hipStream_t)stream) others += (e.pending ? e.need : 0.0) > e.rneed ? (e.pending ? e.need : 0.0) : e.rneed;
    }
    (void)hipGetLastError();
    return (int)(others + 0.5);
}
