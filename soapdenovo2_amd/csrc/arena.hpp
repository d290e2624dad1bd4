// arena.hpp -- one device arena per GPU for everything the library allocates in device memory.
//
// Why: the driver clears what a process releases (27 - 33 GB/s on the boxes measured, scripts/startup_probe.hip) and an allocation that lands
// on memory still being cleared waits for it -- a command that makes dozens of hipMalloc / hipFree calls of tens of gigabytes (pass 1's pool,
// the export array, the set layout, the tip / edge / pass-2 temporaries) waits behind its OWN hipFrees: 0.7 s became 3.65 s of "pass 2 batches"
// on one box with nothing of it in the kernels.  Here a GPU's memory is ONE reserved virtual range (hipMemAddressReserve) backed
// by physical chunks (hipMemCreate + hipMemMap) as its high-water mark grows; blocks are cut from it first-fit and go back to a free list --
// never to the driver -- until the arena is empty and nobody pins it.  call_pregraph pins its devices for the whole command, so the command's
// physical memory is created once, on the way up, and released once, at its end.
//
// A range lives from its first block to the moment the arena is empty and unpinned; its pieces are then given back and the RANGE IS RETIRED -- the
// arena's next life gets a fresh reservation (mapping new memory at addresses the GPU already had translations for read back wrong data,
// profiles/r05w_arena_remap_hazard.txt).  A retired range costs address space only (288 GB of 2^47 each); a process that has used that up -- hundreds
// of create / destroy cycles without a pin -- goes on with plain hipMalloc.  Callers that cycle contexts pin the device around the loop (ArenaPin,
// pg_device_arena_pin): one life, one range.
//
// arena_malloc / arena_free have hipMalloc's / hipFree's contract (current device; arena_free waits for the device like hipFree does, so a block
// is never handed out again under a kernel that still uses it).  A GPU or driver without the virtual-memory calls, or SOAPDENOVO2_AMD_ARENA=0,
// makes both plain hipMalloc / hipFree.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stddef.h>
#include <stdint.h>

namespace pg {

hipError_t arena_malloc(void** p, size_t bytes);
template <typename T> inline hipError_t arena_malloc(T** p, size_t bytes) { return arena_malloc((void**)p, bytes); }
hipError_t arena_free(void* p);
// the block keeps its first `bytes` bytes, the rest goes back to the arena's free list at once (nothing is copied; a block that did not come from
// the arena stays as it is).  The caller knows that nothing in flight touches the part given up.
void arena_shrink(void* p, size_t bytes);

// the arena of `device` keeps its physical memory while it is pinned, even when nothing is allocated from it
void arena_pin(int device);
void arena_unpin(int device);            // (the last unpin of an empty arena releases its physical memory)

// hipMemGetInfo of the current device, with what the arena holds mapped but has not handed out counted as free (it is, to this library)
hipError_t arena_mem_info(size_t* free_bytes, size_t* total_bytes);

struct ArenaStats {
    int active;                          // 1 = the virtual-memory arena, 0 = plain hipMalloc / hipFree
    uint64_t reserved, mapped, in_use, peak_in_use;
    uint64_t n_malloc, n_free, n_chunks_created;
    double map_seconds;                  // spent creating + mapping physical chunks (the driver's clearing shows up here, once)
};
ArenaStats arena_stats(int device);

void arena_pin_for_process(int device);   // pins the device's arena unless somebody holds a pin already (pg_create: the API path)
struct ArenaPin {                        // RAII for call_pregraph
    int device;
    explicit ArenaPin(int d) : device(d) { arena_pin(d); }
    ~ArenaPin() { arena_unpin(device); }
    ArenaPin(const ArenaPin&) = delete;
    ArenaPin& operator=(const ArenaPin&) = delete;
};

}  // namespace pg
