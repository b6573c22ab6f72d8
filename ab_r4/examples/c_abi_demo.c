/* c_abi_demo.c -- libmarinenav_hip.so from plain C: no Python, no torch.
 *
 *   gcc -D__HIP_PLATFORM_AMD__ examples/c_abi_demo.c -Iinclude -I/opt/rocm/include \
 *       -Ldistributional_rl_navigation_amd -lmarinenav_hip -L/opt/rocm/lib -lamdhip64 \
 *       -Wl,-rpath,$PWD/distributional_rl_navigation_amd -Wl,-rpath,/opt/rocm/lib -o /tmp/c_abi_demo
 *   /tmp/c_abi_demo 65536 200
 * (the HIP runtime is used only to allocate the caller-owned buffers)
 *
 * Creates n environments (MarineNavEnv.__init__, marinenav_env.py:27-73), resets them, then runs `steps`
 * vector steps with a fixed action pattern and caller-side reset on done (agent.py:152-170), all device
 * buffers owned by the caller (hipMalloc here; torch tensors in the Python binding).
 */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#include "marinenav_hip.h"

#define CHECK(x) do { int rc_ = (x); if (rc_) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, mn_last_error(h)); return 1; } } while (0)

int main(int argc, char **argv) {
    int n = argc > 1 ? atoi(argv[1]) : 4096, steps = argc > 2 ? atoi(argv[2]) : 100;
    mn_handle *h = NULL;
    mn_params p;
    mn_default_params(&p);
    p.num_cores = 8; p.num_obs = 10; p.min_start_goal_dis = 40.0;     /* curriculum stage 2, train_IQN_model.py:86-90 */
    CHECK(mn_create(n, &p, &h));

    float *obs, *reward; unsigned char *done, *info; int *actions;
    hipMalloc((void **)&obs, (size_t)n * MN_OBS_DIM * sizeof(float));
    hipMalloc((void **)&reward, (size_t)n * sizeof(float));
    hipMalloc((void **)&done, n); hipMalloc((void **)&info, n);
    hipMalloc((void **)&actions, (size_t)n * sizeof(int));
    int *a_host = (int *)malloc((size_t)n * sizeof(int));
    for (int i = 0; i < n; ++i) a_host[i] = (i * 7 + 3) % MN_NUM_ACTIONS;
    hipMemcpy(actions, a_host, (size_t)n * sizeof(int), hipMemcpyHostToDevice);

    CHECK(mn_reset(h, NULL, obs, NULL));
    hipDeviceSynchronize();
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    long finished = 0;
    for (int t = 0; t < steps; ++t) {
        CHECK(mn_step(h, actions, obs, reward, done, info, NULL));
        CHECK(mn_reset_done(h, obs, NULL));
        if (t % 50 == 49) { int c; CHECK(mn_last_done_count(h, NULL, &c)); finished += c; }
    }
    hipDeviceSynchronize();
    clock_gettime(CLOCK_MONOTONIC, &t1);
    double dt = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);

    float *o_host = (float *)malloc(MN_OBS_DIM * sizeof(float));
    hipMemcpy(o_host, obs, MN_OBS_DIM * sizeof(float), hipMemcpyDeviceToHost);
    double st[6]; int ep; long long tot;
    CHECK(mn_get_state(h, 0, 1, st, &ep, (int64_t *)&tot));
    printf("envs %d steps %d: %.1f M env steps/s; env 0: x %.3f y %.3f theta %.3f, episode_timesteps %d, total_timesteps %lld, "
           "goal in robot frame (%.2f, %.2f); sampled done counts %ld\n",
           n, steps, 1e-6 * (double)n * steps / dt, st[0], st[1], st[2], ep, tot, o_host[2], o_host[3], finished);
    if (tot != steps) { fprintf(stderr, "counter mismatch\n"); return 2; }
    mn_destroy(h);
    return 0;
}
