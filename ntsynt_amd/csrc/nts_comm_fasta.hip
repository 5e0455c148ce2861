// libntsynt_hip.so, second translation unit: the two multi-GPU exchanges (nts_comm.inc: AND all-reduce of the filters, all-gather of
// the minimizer lists, over RCCL loaded at run time) and the FASTA parse on the GPU (nts_fasta_dev.inc).  Shared state and helpers:
// nts_internal.h.
#include "nts_internal.h"

#include "nts_comm.inc"
#include "nts_fasta_dev.inc"
