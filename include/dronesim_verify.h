/*
 * dronesim_verify.h -- C ABI of libdronesim_verify.so: the float64 VERIFICATION variant of the env step.
 *
 * TEST INFRASTRUCTURE of the float32 product kernels (include/dronesim.h, libdronesim.so), built as a library of its
 * own so that the product library exports the product only.  Same conventions as dronesim.h (caller-owned device
 * buffers, work enqueued on `stream`, 0 / negative DRONESIM_E* return codes); its own thread-local error string.
 */
#ifndef DRONESIM_VERIFY_H
#define DRONESIM_VERIFY_H

#include <stdint.h>
#include "dronesim.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- float64 VERIFICATION variant (test infrastructure of the float32 product kernels; csrc/verify_f64.hip) -------
 * The reference computes in float64 (drone_env.py:189).  dronesim_step_f64 / dronesim_observe_f64 run the same
 * per-pair arithmetic (one scalar-type template, instantiated for double) and epilogue semantics as dronesim_step /
 * dronesim_observe on float64 buffers -- one workgroup per env, every ordered pair visited, no far filter: slow and
 * simple.  They exist so that the parity tests can (1) meet the reference's golden vectors with no float32-state
 * allowance, (2) follow a free-running 200-step episode of the float64 oracle, (3) judge the float32 kernels against
 * a float64 evaluation of the same state on the device.  Layouts as in dronesim_step with double instead of float.  */
typedef struct DroneParamsF64 {
    int32_t N;
    int32_t k;
    int32_t c;
    int32_t max_steps;
    double dt;
    double q;
    double b;
    double done_radius;
    double ghost_factor;
    const double *xF;       /* [N][2] */
    const double *d_hat;    /* [N]    */
    const double *delta;    /* [N]    */
    const double *radius;   /* [N]    */
} DroneParamsF64;
int dronesim_step_f64(const DroneParamsF64 *p, double *pos, double *vel, int32_t *t, const double *act,
                      double *reward, double *true_reward, double *z, int32_t *nbr_idx,
                      int32_t *n_coll, uint8_t *done, int E, void *stream);
int dronesim_observe_f64(const DroneParamsF64 *p, const double *pos, const double *vel,
                         double *reward, double *true_reward, double *z, int32_t *nbr_idx,
                         int32_t *n_coll, int E, void *stream);

/* thread-local description of the last failure of the two entry points above */
const char *dronesim_verify_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* DRONESIM_VERIFY_H */
