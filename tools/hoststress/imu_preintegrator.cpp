// Host-side stress load of BASELINE.json configs[4] ("matcher + IMU-preintegration stress"): IMUPreintegrator::update
// (src/IMU/IMUPreintegrator.cpp:81-140; helpers src/IMU/IMUPreintegrator.h:80-178) restated in plain C++ -- no Eigen, no Sophus.
// It runs on the tracking thread between the frames of the configs[4] loop exactly where the reference runs it (10 samples per
// frame at 200 Hz / 20 Hz); it is NOT part of the GPU hot path and is not a parity target: the rotation update uses Rodrigues'
// formula directly where the reference goes through Sophus' quaternion exponential, and the re-normalisation goes through a
// quaternion as normalizeRotationM does.  tests/test_host_logic.py checks it against closed forms.
//   g++ -O3 -shared -fPIC -o libimu_stress.so imu_preintegrator.cpp
#include <cmath>
#include <cstring>

namespace {

struct M3 {
  double a[9];
};
inline M3 ident() { return M3{{1, 0, 0, 0, 1, 0, 0, 0, 1}}; }
inline M3 zero3() { return M3{{0, 0, 0, 0, 0, 0, 0, 0, 0}}; }
inline M3 mul(const M3& x, const M3& y) {
  M3 r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r.a[3 * i + j] = x.a[3 * i] * y.a[j] + x.a[3 * i + 1] * y.a[3 + j] + x.a[3 * i + 2] * y.a[6 + j];
  return r;
}
inline M3 tr(const M3& x) { return M3{{x.a[0], x.a[3], x.a[6], x.a[1], x.a[4], x.a[7], x.a[2], x.a[5], x.a[8]}}; }
inline M3 scale(const M3& x, double s) {
  M3 r;
  for (int i = 0; i < 9; ++i) r.a[i] = x.a[i] * s;
  return r;
}
inline M3 add(const M3& x, const M3& y) {
  M3 r;
  for (int i = 0; i < 9; ++i) r.a[i] = x.a[i] + y.a[i];
  return r;
}
inline M3 skew(const double* v) { return M3{{0, -v[2], v[1], v[2], 0, -v[0], -v[1], v[0], 0}}; }  // SO3::hat
inline void mulv(const M3& x, const double* v, double* o) {
  for (int i = 0; i < 3; ++i) o[i] = x.a[3 * i] * v[0] + x.a[3 * i + 1] * v[1] + x.a[3 * i + 2] * v[2];
}
// Expmap: Rodrigues (IMUPreintegrator.h:87-90)
M3 expmap(const double* w) {
  const double th = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  const M3 W = skew(w);
  if (th < 1e-10) return add(ident(), W);
  return add(add(ident(), scale(W, std::sin(th) / th)), scale(mul(W, W), (1 - std::cos(th)) / (th * th)));
}
// JacobianR (IMUPreintegrator.h:93-110)
M3 jacobian_r(const double* w) {
  const double th = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  if (th < 0.00001) return ident();
  const double k[3] = {w[0] / th, w[1] / th, w[2] / th};
  const M3 K = skew(k);
  return add(add(ident(), scale(K, -(1 - std::cos(th)) / th)), scale(mul(K, K), 1 - std::sin(th) / th));
}
// normalizeRotationM (IMUPreintegrator.h:164-178): matrix -> quaternion (w >= 0) -> normalised -> matrix
M3 normalize_rotation(const M3& R) {
  double q[4];  // w x y z, Eigen's conversion (Shepperd)
  const double t = R.a[0] + R.a[4] + R.a[8];
  if (t > 0) {
    double s = std::sqrt(t + 1.0);
    q[0] = 0.5 * s;
    s = 0.5 / s;
    q[1] = (R.a[7] - R.a[5]) * s, q[2] = (R.a[2] - R.a[6]) * s, q[3] = (R.a[3] - R.a[1]) * s;
  } else {
    int i = 0;
    if (R.a[4] > R.a[0]) i = 1;
    if (R.a[8] > R.a[4 * i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    double s = std::sqrt(R.a[4 * i] - R.a[4 * j] - R.a[4 * k] + 1.0);
    q[1 + i] = 0.5 * s;
    s = 0.5 / s;
    q[0] = (R.a[3 * k + j] - R.a[3 * j + k]) * s;
    q[1 + j] = (R.a[3 * j + i] + R.a[3 * i + j]) * s;
    q[1 + k] = (R.a[3 * k + i] + R.a[3 * i + k]) * s;
  }
  if (q[0] < 0)
    for (double& v : q) v = -v;
  const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  const double w = q[0] / n, x = q[1] / n, y = q[2] / n, z = q[3] / n;
  return M3{{1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y), 2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
             2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)}};
}

struct Preint {
  double dP[3], dV[3], dt_sum;
  M3 dR, JPg, JPa, JVg, JVa, JRg;
  double cov[81];
  void reset() {
    std::memset(this, 0, sizeof(*this));
    dR = ident();
  }
  // update(omega, acc, dt): :81-140
  void update(const double* omega, const double* acc, double dt, double gyr_cov, double acc_cov) {
    const double dt2 = dt * dt;
    const double wdt[3] = {omega[0] * dt, omega[1] * dt, omega[2] * dt};
    const M3 dRk = expmap(wdt), Jr = jacobian_r(wdt), Sa = skew(acc);
    // err_k+1 = A err_k + Bg err_gyro + Ca err_acc  (9x9, blocks of 3)
    double A[81];
    for (int i = 0; i < 81; ++i) A[i] = (i % 10 == 0) ? 1.0 : 0.0;
    const M3 dRt = tr(dRk), RS = mul(dR, Sa);
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        A[(6 + i) * 9 + 6 + j] = dRt.a[3 * i + j];
        A[(3 + i) * 9 + 6 + j] = -RS.a[3 * i + j] * dt;
        A[i * 9 + 6 + j] = -0.5 * RS.a[3 * i + j] * dt2;
        A[i * 9 + 3 + j] = (i == j) ? dt : 0.0;
      }
    double T[81], N[81];
    for (int i = 0; i < 9; ++i)
      for (int j = 0; j < 9; ++j) {
        double s = 0;
        for (int k = 0; k < 9; ++k) s += A[i * 9 + k] * cov[k * 9 + j];
        T[i * 9 + j] = s;
      }
    for (int i = 0; i < 9; ++i)
      for (int j = 0; j < 9; ++j) {
        double s = 0;
        for (int k = 0; k < 9; ++k) s += T[i * 9 + k] * A[j * 9 + k];
        N[i * 9 + j] = s;
      }
    // Bg = [0; 0; Jr dt], Ca = [0.5 dR dt2; dR dt; 0] with isotropic measurement covariances
    const M3 JJ = scale(mul(Jr, tr(Jr)), dt * dt * gyr_cov), RR = mul(dR, tr(dR));
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        N[(6 + i) * 9 + 6 + j] += JJ.a[3 * i + j];
        N[i * 9 + j] += 0.25 * dt2 * dt2 * acc_cov * RR.a[3 * i + j];
        N[i * 9 + 3 + j] += 0.5 * dt2 * dt * acc_cov * RR.a[3 * i + j];
        N[(3 + i) * 9 + j] += 0.5 * dt2 * dt * acc_cov * RR.a[3 * i + j];
        N[(3 + i) * 9 + 3 + j] += dt2 * acc_cov * RR.a[3 * i + j];
      }
    std::memcpy(cov, N, sizeof(cov));
    // jacobians w.r.t. the biases: P first, then V, then R
    const M3 RSJ = mul(RS, JRg);
    JPa = add(JPa, add(scale(JVa, dt), scale(dR, -0.5 * dt2)));
    JPg = add(JPg, add(scale(JVg, dt), scale(RSJ, -0.5 * dt2)));
    JVa = add(JVa, scale(dR, -dt));
    JVg = add(JVg, scale(RSJ, -dt));
    JRg = add(mul(dRt, JRg), scale(Jr, -dt));
    // delta measurements: P first (needs the previous V and R), then V, then R
    double Ra[3];
    mulv(dR, acc, Ra);
    for (int i = 0; i < 3; ++i) dP[i] += dV[i] * dt + 0.5 * Ra[i] * dt2;
    for (int i = 0; i < 3; ++i) dV[i] += Ra[i] * dt;
    dR = normalize_rotation(mul(dR, dRk));
    dt_sum += dt;
  }
};

}  // namespace

// samples: [n][7] = wx wy wz ax ay az dt (bias already removed).  out[17]: delta_P[3], delta_V[3], delta_R[9] row-major, delta_time,
// trace of the 9x9 covariance.  reset_every > 0: the preintegrator is reset every that many samples (one frame's worth); the outputs
// are those of the last segment.
extern "C" void imu_preintegrate(const double* samples, int n, int reset_every, double gyr_cov, double acc_cov, double* out) {
  Preint P;
  P.reset();
  for (int i = 0; i < n; ++i) {
    if (reset_every > 0 && i % reset_every == 0) P.reset();
    P.update(samples + 7 * i, samples + 7 * i + 3, samples[7 * i + 6], gyr_cov, acc_cov);
  }
  std::memcpy(out, P.dP, 24);
  std::memcpy(out + 3, P.dV, 24);
  std::memcpy(out + 6, P.dR.a, 72);
  out[15] = P.dt_sum;
  double tr9 = 0;
  for (int i = 0; i < 9; ++i) tr9 += P.cov[i * 10];
  out[16] = tr9;
}
