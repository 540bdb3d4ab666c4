/* ORACLE (test infrastructure only): see fac_ref_impl.h for the reference citations. */
#include <stdint.h>

#define REAL float
#define SUFFIX _f32
#include "fac_ref_impl.h"
#undef REAL
#undef SUFFIX

#define REAL double
#define SUFFIX _f64
#include "fac_ref_impl.h"
#undef REAL
#undef SUFFIX
