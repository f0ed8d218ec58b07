// Synthetic stand-in for the reference's PyBullet Environment, E independent arms stepped on the device.
// NOT a port of Bullet: a kinematic serial chain (alternating z / y revolute axes, velocity control applied
// exactly) that keeps the reference's observable contract so the hot path sees data of the right shape:
//   state  = [q(A), qdot(A), end-effector xyz, target xyz, obstacle xyz]   (environment.py:431-451, S = 9+2A)
//   reward = +250 on reaching the target (dist < 0.05), -1000 on obstacle contact, else -(dist - 0.05)
//            (environment.py:345-371, :419-429)
//   done   = 1 on either terminal event (environment.py:311-333)
//   step   = velocity control for one 1/240 s simulation tick (environment.py:453-485)
// PyBullet is not installed in the image and has no pinned version upstream: env parity is "unpinned";
// this file exists so env-steps/s can be measured without leaving the GPU.
#include "common.h"
#include "../../include/naf_hip.h"

// q[8] | target[3] | obstacle[3] | init_q[8] | frame | episode | variation[8] | episode score (one double) | pad[2]
// Per-env episode bookkeeping (the reference's `score += reward` / `frame` of NAFAgent.run, naf_algorithm.py:246-270)
// rides in the step kernel: the running score is a DOUBLE summed in step order, as Python's float sum is.
#define ENV_STATE_FLOATS 36
#define ENV_OFF_VAR 24
#define ENV_OFF_SCORE 32
#define ENV_DT (1.0f / 240.0f)

struct EnvCfg {
    float link[8];
};
__constant__ EnvCfg c_env = {{0.34f, 0.02f, 0.40f, 0.02f, 0.40f, 0.13f, 0.05f, 0.05f}};

__device__ static inline void fk_chain(const float* q, int A, float* ee, const float* obstacle, float obstacle_r,
                                       bool* hit) {
    // R accumulates the orientation, p the position of the current joint frame
    float R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    float p[3] = {0.f, 0.f, 0.f};
    bool h = false;
    for (int k = 0; k < A; ++k) {
        float c = cosf(q[k]), s = sinf(q[k]);
        float N[9];
        if ((k & 1) == 0) {  // about z:  R * Rz
            for (int r = 0; r < 3; ++r) {
                N[r * 3 + 0] = R[r * 3 + 0] * c + R[r * 3 + 1] * s;
                N[r * 3 + 1] = -R[r * 3 + 0] * s + R[r * 3 + 1] * c;
                N[r * 3 + 2] = R[r * 3 + 2];
            }
        } else {  // about y:  R * Ry
            for (int r = 0; r < 3; ++r) {
                N[r * 3 + 0] = R[r * 3 + 0] * c - R[r * 3 + 2] * s;
                N[r * 3 + 1] = R[r * 3 + 1];
                N[r * 3 + 2] = R[r * 3 + 0] * s + R[r * 3 + 2] * c;
            }
        }
        for (int e = 0; e < 9; ++e) R[e] = N[e];
        const float l = c_env.link[k];
        p[0] += R[2] * l; p[1] += R[5] * l; p[2] += R[8] * l;
        float dx = p[0] - obstacle[0], dy = p[1] - obstacle[1], dz = p[2] - obstacle[2];
        h |= (dx * dx + dy * dy + dz * dz) < obstacle_r * obstacle_r;
    }
    ee[0] = p[0]; ee[1] = p[1]; ee[2] = p[2];
    *hit = h;
}

__device__ static inline void write_obs(float* o, const float* q, const float* qd, const float* ee, const float* target,
                                        const float* obstacle, int A) {
    for (int k = 0; k < A; ++k) { o[k] = q[k]; o[A + k] = qd[k]; }
    for (int k = 0; k < 3; ++k) { o[2 * A + k] = ee[k]; o[2 * A + 3 + k] = target[k]; o[2 * A + 6 + k] = obstacle[k]; }
}

__device__ static inline void env_reset_one(float* st, int e, int A, uint64_t seed, uint64_t ctr) {
    // initial joint positions + uniform(-0.1, 0.1) variation (environment.py:284-293 semantics)
    for (int k = 0; k < A; k += 4) {
        Philox4 p = philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), (uint32_t)e, 0x52455345u + k, (uint32_t)seed,
                                  (uint32_t)(seed >> 32));
        for (int j = 0; j < 4 && k + j < A; ++j)
            st[k + j] = st[14 + k + j] + (naf_u01(p.v[j]) * 2.f - 1.f) * st[ENV_OFF_VAR + k + j];
    }
    st[22] = 0.f;
    *(double*)(st + ENV_OFF_SCORE) = 0.0;
}

// preset: [init_q(8) | target(3) | obstacle(3) | obstacle_jitter | variation(8)] — the demo presets of the reference
// (rl_framework.py:547-555 KUKA, :571-580 xArm6) or the caller's own; obstacle_jitter > 0 gives every env its own
// obstacle position, uniform in a cube of that half-width around the preset (BASELINE configs[3]); variation[k] = half-width
// of the uniform range joint k's initial position is drawn from at every reset (initial_positions_variation_range,
// environment.py:284-293)
struct EnvPreset {
    float v[NAF_SYNTH_PRESET_FLOATS];
};

__global__ void synth_env_reset_kernel(float* env_state, float* obs, int E, int A, uint64_t seed, uint64_t ctr,
                                       const EnvPreset preset) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    float* st = env_state + (int64_t)e * ENV_STATE_FLOATS;
    for (int k = 0; k < 8; ++k) { st[14 + k] = preset.v[k]; st[ENV_OFF_VAR + k] = preset.v[15 + k]; }
    st[34] = st[35] = 0.f;
    for (int k = 0; k < 3; ++k) { st[8 + k] = preset.v[8 + k]; st[11 + k] = preset.v[11 + k]; }
    if (preset.v[14] > 0.f) {
        Philox4 p = philox4x32_10((uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)e, 0x4f425354u, 0x9E3779B9u, 0x243F6A88u);
        for (int k = 0; k < 3; ++k) st[11 + k] += (naf_u01(p.v[k]) * 2.f - 1.f) * preset.v[14];
    }
    st[23] = 0.f;
    env_reset_one(st, e, A, seed, ctr);
    float qd[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ee[3];
    bool hit;
    fk_chain(st, A, ee, st + 11, 0.06f, &hit);
    write_obs(obs + (int64_t)e * (2 * A + 9), st, qd, ee, st + 8, st + 11, A);
}

__global__ void synth_env_step_kernel(float* env_state, const float* __restrict__ actions, float* __restrict__ out_rows,
                                      float* __restrict__ obs_next, int E, int A, int row_floats, uint64_t seed,
                                      const uint64_t* counter_dev, int max_frames, naf_episode_record_t* __restrict__ records,
                                      int record_slots) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const int S = 2 * A + 9;
    float* st = env_state + (int64_t)e * ENV_STATE_FLOATS;
    float* row = out_rows + (int64_t)e * row_floats;
    float* ob = obs_next + (int64_t)e * S;
    const uint64_t ctr = counter_dev ? *counter_dev : 0ull;

    // the observation the action was chosen from is the row's `state`
    for (int k = 0; k < S; ++k) row[k] = ob[k];
    float a[8], ee[3];
    for (int k = 0; k < A; ++k) {
        a[k] = actions[(int64_t)e * A + k];
        row[S + k] = a[k];
        st[k] += ENV_DT * a[k];  // velocity control: the commanded velocity is reached within the tick
    }
    bool hit;
    fk_chain(st, A, ee, st + 11, 0.06f, &hit);
    float dx = ee[0] - st[8], dy = ee[1] - st[9], dz = ee[2] - st[10];
    float dist = sqrtf(dx * dx + dy * dy + dz * dz);
    const bool reached = dist < 0.05f;
    float reward = reached ? 250.f : (hit ? -1000.f : -(dist - 0.05f));
    float done = (reached || hit) ? 1.f : 0.f;
    const int off_s2 = naf_row_off_s2(S, A), off_d = naf_row_off_done(S, A);
    row[S + A] = reward;
    for (int k = S + A + 1; k < off_s2; ++k) row[k] = 0.f;
    write_obs(row + off_s2, st, a, ee, st + 8, st + 11, A);
    row[off_d] = done;
    for (int k = off_d + 1; k < row_floats; ++k) row[k] = 0.f;

    st[22] += 1.f;
    const double score = *(double*)(st + ENV_OFF_SCORE) + (double)reward;      // score += reward (naf_algorithm.py:264)
    *(double*)(st + ENV_OFF_SCORE) = score;
    const bool over = done != 0.f || (max_frames > 0 && st[22] >= (float)max_frames);
    if (records) {
        // one record slot per (vector step mod record_slots, env), written EVERY step (frames == 0: no episode ended here):
        // the host drains whole slots in (step, env) order — completion order without an atomic or a per-step sync
        naf_episode_record_t rec;
        rec.score = over ? score : 0.0;
        rec.frames = over ? (int32_t)st[22] : 0;
        rec.done = (int32_t)done;
        rec.last_reward = reward;
        rec.episode = (int32_t)st[23] + 1;
        rec.step_lo = (uint32_t)ctr;
        rec.env = (uint32_t)e;
        records[(int64_t)(ctr % (uint64_t)record_slots) * E + e] = rec;
    }
    if (over) {
        // episode over (terminal state, or the frame budget of NAFAgent.run, naf_algorithm.py:249): auto-reset
        st[23] += 1.f;
        env_reset_one(st, e, A, seed, ctr * 0x9E3779B97F4A7C15ull + (uint64_t)st[23]);
        float qd0[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        fk_chain(st, A, ee, st + 11, 0.06f, &hit);
        write_obs(ob, st, qd0, ee, st + 8, st + 11, A);
    } else {
        for (int k = 0; k < S; ++k) ob[k] = row[off_s2 + k];
    }
}

extern "C" int naf_synth_env_state_floats(int A) { return (A > 0 && A <= NAF_MAX_A) ? ENV_STATE_FLOATS : NAF_ERR_ARG; }

extern "C" int naf_synth_env_reset(float* env_state, float* obs, int E, int A, uint64_t seed, uint64_t counter,
                                   const float* preset_host, int preset_floats, void* stream) {
    if (!env_state || !obs || E <= 0 || A <= 0 || A > NAF_MAX_A) return NAF_ERR_ARG;
    if (preset_host && preset_floats != 15 && preset_floats != NAF_SYNTH_PRESET_FLOATS) return NAF_ERR_ARG;
    // default: the reference's KUKA demo preset (rl_framework.py:551-553), no obstacle jitter, +-0.1 on every joint
    EnvPreset p = {{0.9f, 0.45f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.4f, 0.85f, 0.71f, 0.45f, 0.55f, 0.55f, 0.f,
                    0.1f, 0.1f, 0.1f, 0.1f, 0.1f, 0.1f, 0.1f, 0.1f}};
    if (preset_host)
        for (int k = 0; k < preset_floats; ++k) p.v[k] = preset_host[k];
    synth_env_reset_kernel<<<(E + 63) / 64, 64, 0, (hipStream_t)stream>>>(env_state, obs, E, A, seed, counter, p);
    NAF_CHECK_LAUNCH();
    return NAF_OK;
}

extern "C" int naf_synth_env_step(float* env_state, const float* actions, float* out_rows, float* obs_next, int E, int A,
                                  uint64_t seed, const uint64_t* counter_dev, int max_frames, naf_episode_record_t* records,
                                  int record_slots, void* stream) {
    if (!env_state || !actions || !out_rows || !obs_next || E <= 0 || A <= 0 || A > NAF_MAX_A) return NAF_ERR_ARG;
    if (records && (record_slots <= 0 || !counter_dev)) return NAF_ERR_ARG;
    int rf = naf_replay_row_floats(2 * A + 9, A);
    synth_env_step_kernel<<<(E + 63) / 64, 64, 0, (hipStream_t)stream>>>(env_state, actions, out_rows, obs_next, E, A, rf,
                                                                         seed, counter_dev, max_frames, records, record_slots);
    NAF_CHECK_LAUNCH();
    return NAF_OK;
}
