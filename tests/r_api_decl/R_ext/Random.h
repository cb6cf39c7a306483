/* TEST INFRASTRUCTURE ONLY — see ../Rinternals.h. */
#ifndef S4B_TEST_R_RANDOM_H
#define S4B_TEST_R_RANDOM_H
#ifdef __cplusplus
extern "C" {
#endif
void GetRNGstate(void);
void PutRNGstate(void);
#ifdef __cplusplus
}
#endif
#endif
