/* TEST INFRASTRUCTURE ONLY — see ../Rinternals.h. */
#ifndef S4B_TEST_R_UTILS_H
#define S4B_TEST_R_UTILS_H
#endif
