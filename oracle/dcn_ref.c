/* ORACLE (test infrastructure only): see dcn_ref_impl.h for the reference citations. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define REAL float
#define SUFFIX _f32
#include "dcn_ref_impl.h"
#undef REAL
#undef SUFFIX

#define REAL double
#define SUFFIX _f64
#include "dcn_ref_impl.h"
#undef REAL
#undef SUFFIX
