// Timing-ablation macros (tools/kbench, tools/ab_bench.sh: builds that skip loads / MFMAs / LDS reads to see what a kernel is
// paced by -- their results are WRONG by design) must never reach the product library.  Every translation unit of csrc/
// includes this header through common.hpp: defining any of them without EBFI_ABLATE (set only by tools/build_kbench.sh and by
// A/B builds into a separate output file, csrc/build.sh) stops the compilation.
#pragma once
#if !defined(EBFI_ABLATE) && !defined(EBFI_KBENCH)
#if defined(KB_NO_LOADS) || defined(KB_NO_MFMA) || defined(KB_NO_LDSREAD) || defined(KB_NO_SLAB) || defined(KB_NO_WLOAD) ||      \
    defined(ABL_ONE_MFMA) || defined(WS_NO_PRODUCE) || defined(WS_NO_CONSUME) || defined(WS_NO_LDSREAD) ||                     \
    defined(WSF_NO_PRODUCE) || defined(WSF_NO_CONSUME) || defined(DCN_STAMPS)
#error "a timing-ablation macro is defined in a product build of libebfi_hip.so (results would be wrong): add -DEBFI_ABLATE and build into a separate file with EBFI_LIB_OUT"
#endif
#endif
