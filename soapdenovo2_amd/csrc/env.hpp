// env.hpp -- every environment switch of the library goes through one of the three functions below; nothing else in csrc/ calls getenv.
//
//   env_user(name)     the switches INTEGRATION.md §4 documents for whoever runs the command: device choice, batch sizes, thread counts,
//                      the explicit A/B forms of a stage (host twins), verbosity.
//   env_test(name)     hooks the test-suite uses to force a corner of the product path that real inputs reach only at scale (a send region that
//                      overflows, a partition that outgrows its chunk list, a reader window of a few kilobytes, ...).  Listed in INTEGRATION.md §4
//                      as such; they select among code paths the product has anyway, never a different algorithm.
//   env_measure(name)  geometry and scheduling knobs of the kernels that were swept while they were tuned (tile sizes, partition counts,
//                      workgroups a CU, ...).  The product is built WITHOUT -DPG_MEASURE: there these return nullptr, i.e. every knob is its
//                      default and costs nothing; `make MEASURE=1` builds the library the A/B runs under profiles/ were taken with.
#pragma once
#include <stdlib.h>

namespace pg {

inline const char* env_user(const char* name) { return getenv(name); }
inline const char* env_test(const char* name) { return getenv(name); }
#ifdef PG_MEASURE
inline const char* env_measure(const char* name) { return getenv(name); }
constexpr bool kMeasureBuild = true;
#else
inline const char* env_measure(const char*) { return nullptr; }
constexpr bool kMeasureBuild = false;
#endif

inline int env_int(const char* v, int dflt) { return v && *v ? atoi(v) : dflt; }
inline bool env_on(const char* v) { return v && atoi(v) != 0; }

}  // namespace pg
