/* TEST INFRASTRUCTURE ONLY — see Rinternals.h in this directory. */
#ifndef S4B_TEST_R_H
#define S4B_TEST_R_H
#include "Rinternals.h"
#include "R_ext/Random.h"
#include "R_ext/Utils.h"
#endif
