// Host side of the -DMSK144_PHASE_STAMPS diagnostic build (phase_stamps.h): the stamp buffers and the entry tools/phase_stamps.py
// reads them through.  Not part of the product library (build.py compiles it only with stamps=True) and not in include/msk144hip.h.
#include "phase_stamps.h"

#ifdef MSK144_PHASE_STAMPS
namespace msk144
{

static uint64_t* g_stamps[2] = {nullptr, nullptr};

uint64_t* stamp_buffer(int kernel)
{
    if(kernel < 0 || kernel > 1) return nullptr;
    if(!g_stamps[kernel])
    {
        const size_t bytes = sizeof(uint64_t) * kStampSlots * kStampRows;
        if(hipMalloc(reinterpret_cast<void**>(&g_stamps[kernel]), bytes) != hipSuccess) return nullptr;
        (void)hipMemset(g_stamps[kernel], 0, bytes);
    }
    return g_stamps[kernel];
}

}  // namespace msk144

// rows x slots uint64, copied after a device synchronisation, then zeroed; returns the number of rows copied (<= max_rows)
extern "C" int msk144_debug_read_stamps(int kernel, uint64_t* out, int max_rows, int* slots, int* every)
{
    using namespace msk144;
    uint64_t* buf = stamp_buffer(kernel);
    if(!buf || !out) return -1;
    if(hipDeviceSynchronize() != hipSuccess) return -2;
    const int rows = max_rows < kStampRows ? max_rows : kStampRows;
    if(hipMemcpy(out, buf, sizeof(uint64_t) * kStampSlots * rows, hipMemcpyDeviceToHost) != hipSuccess) return -2;
    (void)hipMemset(buf, 0, sizeof(uint64_t) * kStampSlots * kStampRows);
    if(slots) *slots = kStampSlots;
    if(every) *every = kStampEvery;
    return rows;
}
#endif
