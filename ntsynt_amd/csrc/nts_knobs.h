// Experiment switches of the library.  The environment variables that pick a kernel variant, force a fallback, cut a list short
// or (NTS_ACC_NO_LOOKUP) change a result for a timing experiment exist only in the NTS_EXPERIMENTS build
// (ntsynt_amd/libntsynt_hip_exp.so: the tests and scripts that use them load that one, ntsynt_amd/_lib.py `variant`); in the
// product build NTS_KNOB(...) is a null pointer, the branches behind it fold away and the names are not in the binary.
// What the product build does read from the environment, once per context (nts_init) or process: NTS_RCCL_LIB (which librccl),
// NTS_COMM_PIECE (bytes per piece of the reduce-scatter), NTS_IO_THREADS, NTS_HOST_THREADS (host threads of the file transfers
// and of the host-side passes).
#pragma once
#include <cstdlib>
#ifdef NTS_EXPERIMENTS
#define NTS_KNOB(name) getenv(name)
#else
inline const char* nts_knob_off() { return nullptr; }
#define NTS_KNOB(name) nts_knob_off()
#endif
