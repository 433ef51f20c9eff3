/* Stand-in for R's <R.h> (see Rinternals.h in this directory): the shim needs nothing from it beyond what
 * Rinternals.h declares. */
#ifndef INSIDER_STUB_R_H
#define INSIDER_STUB_R_H
#include <stdlib.h>
#include <stdio.h>
#endif
