// vector.hh — small fixed-size vectors usable as pixel types (reference: vpp/core/vector.hh:10-109, which aliases
// Eigen::Matrix<T,N,1>; this is a self-contained POD with the subset of that interface the vpp API exposes).
// Layout: exactly N * sizeof(T) bytes (sizeof(vuchar3) == 3, sizeof(vfloat2) == 8) — pitches depend on it.
#pragma once
#include <cmath>
#include <type_traits>

// In a translation unit compiled by hipcc the pixel types are also usable inside device code (pixel_wise lambdas evaluated on
// the GPU, vpp/core/pixel_wise_device.hh): every member below is host + device there.
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define VPP_HD __host__ __device__
#else
#define VPP_HD
#endif

// The reference's pixel vectors ARE Eigen::Matrix<T,N,1> (vpp/core/vector.hh:10-20), and user code may name their Eigen bases
// (tests/cast.cc:9-13 binds a vuchar3 to `const Eigen::MatrixBase<vuchar3>&` and asks is_base_of<Eigen::EigenBase<V>, V>).  The two CRTP
// tags below are empty (the vector keeps its exact N * sizeof(T) layout through the empty-base optimisation) and give back the
// vector through derived(); Eigen::Matrix<T,N,1> names the same type as vpp::vector<T,N>.
namespace Eigen {
template <class D> struct EigenBase {
  VPP_HD const D& derived() const { return *static_cast<const D*>(this); }
  VPP_HD D& derived() { return *static_cast<D*>(this); }
};
template <class D> struct MatrixBase : EigenBase<D> {
  // a vector compared with ITSELF through its base (tests/cast.cc:10-13 does so with an uninitialised vuchar3) is equal by identity for integer
  // components — no read of the indeterminate bytes, which an optimiser may otherwise treat as two different values — and by value (NaN != NaN) for floats
  static VPP_HD bool same(const D& a, const D& b) {
    if (&a == &b && std::is_integral<typename D::Scalar>::value) return true;
    return a == b;
  }
  friend VPP_HD bool operator==(const D& a, const MatrixBase& b) { return same(a, b.derived()); }
  friend VPP_HD bool operator==(const MatrixBase& a, const D& b) { return same(a.derived(), b); }
  friend VPP_HD bool operator!=(const D& a, const MatrixBase& b) { return !same(a, b.derived()); }
  friend VPP_HD bool operator!=(const MatrixBase& a, const D& b) { return !same(a.derived(), b); }
};
}  // namespace Eigen

namespace vpp {

template <class T, unsigned N> struct vector : Eigen::MatrixBase<vector<T, N>> {
  typedef T Scalar;
  enum { SizeAtCompileTime = N };
  T v[N];

  vector() = default;
  template <class A, unsigned M = N, class = typename std::enable_if<M == 1 && std::is_arithmetic<A>::value>::type> VPP_HD explicit vector(A a) { v[0] = T(a); }   // vfloat1(1): Eigen's 1x1 scalar constructor
  template <class A, class B, unsigned M = N, class = typename std::enable_if<M == 2>::type> VPP_HD vector(A a, B b) { v[0] = T(a); v[1] = T(b); }
  template <class A, class B, class C, unsigned M = N, class = typename std::enable_if<M == 3>::type> VPP_HD vector(A a, B b, C c) { v[0] = T(a); v[1] = T(b); v[2] = T(c); }
  template <class A, class B, class C, class D, unsigned M = N, class = typename std::enable_if<M == 4>::type> VPP_HD vector(A a, B b, C c, D d) { v[0] = T(a); v[1] = T(b); v[2] = T(c); v[3] = T(d); }
  VPP_HD explicit vector(const T* p) { for (unsigned i = 0; i < N; i++) v[i] = p[i]; }

  VPP_HD static vector Zero() { vector r; for (unsigned i = 0; i < N; i++) r.v[i] = T(0); return r; }
  VPP_HD static vector Ones() { vector r; for (unsigned i = 0; i < N; i++) r.v[i] = T(1); return r; }
  VPP_HD static constexpr int size() { return N; }
  VPP_HD T& operator[](int i) { return v[i]; }
  VPP_HD const T& operator[](int i) const { return v[i]; }
  VPP_HD T& operator()(int i) { return v[i]; }
  VPP_HD const T& operator()(int i) const { return v[i]; }
  template <class U> VPP_HD vector<U, N> cast() const { vector<U, N> r; for (unsigned i = 0; i < N; i++) r.v[i] = U(v[i]); return r; }
  template <unsigned K> VPP_HD vector<T, K> segment(int start) const { vector<T, K> r; for (unsigned i = 0; i < K; i++) r.v[i] = v[start + i]; return r; }

  VPP_HD vector operator-() const { vector r; for (unsigned i = 0; i < N; i++) r.v[i] = -v[i]; return r; }
  VPP_HD vector& operator+=(const vector& o) { for (unsigned i = 0; i < N; i++) v[i] += o.v[i]; return *this; }
  VPP_HD vector& operator-=(const vector& o) { for (unsigned i = 0; i < N; i++) v[i] -= o.v[i]; return *this; }
  template <class S, class = typename std::enable_if<std::is_arithmetic<S>::value>::type> VPP_HD vector& operator*=(S s) { for (unsigned i = 0; i < N; i++) v[i] *= T(s); return *this; }
  template <class S, class = typename std::enable_if<std::is_arithmetic<S>::value>::type> VPP_HD vector& operator/=(S s) { for (unsigned i = 0; i < N; i++) v[i] /= T(s); return *this; }
  VPP_HD bool operator==(const vector& o) const { for (unsigned i = 0; i < N; i++) if (!(v[i] == o.v[i])) return false; return true; }
  VPP_HD bool operator!=(const vector& o) const { return !(*this == o); }
  VPP_HD T squaredNorm() const { T s = v[0] * v[0]; for (unsigned i = 1; i < N; i++) s += v[i] * v[i]; return s; }
  VPP_HD T norm() const { return T(std::sqrt(squaredNorm())); }
  VPP_HD T dot(const vector& o) const { T s = v[0] * o.v[0]; for (unsigned i = 1; i < N; i++) s += v[i] * o.v[i]; return s; }

  // `m << a, b, c;` (Eigen's comma initialiser, the spelling the reference's tests use for small vectors): fills the components in order
  struct comma_filler {
    vector* self; unsigned next;
    template <class S> VPP_HD comma_filler& operator,(S s) { if (next < N) self->v[next++] = T(s); return *this; }
  };
  template <class S, class = typename std::enable_if<std::is_arithmetic<S>::value>::type> VPP_HD comma_filler operator<<(S s) { v[0] = T(s); return comma_filler{this, 1u}; }
};

template <class T, unsigned N> VPP_HD vector<T, N> operator+(vector<T, N> a, const vector<T, N>& b) { a += b; return a; }
template <class T, unsigned N> VPP_HD vector<T, N> operator-(vector<T, N> a, const vector<T, N>& b) { a -= b; return a; }
template <class T, unsigned N, class S, class = typename std::enable_if<std::is_arithmetic<S>::value>::type> VPP_HD vector<T, N> operator*(vector<T, N> a, S s) { a *= s; return a; }
template <class T, unsigned N, class S, class = typename std::enable_if<std::is_arithmetic<S>::value>::type> VPP_HD vector<T, N> operator*(S s, vector<T, N> a) { for (unsigned i = 0; i < N; i++) a.v[i] = T(s) * a.v[i]; return a; }
template <class T, unsigned N, class S, class = typename std::enable_if<std::is_arithmetic<S>::value>::type> VPP_HD vector<T, N> operator/(vector<T, N> a, S s) { a /= s; return a; }

#define VPP_ALIAS_DECL(T1, T2)            \
  template <unsigned N> using v##T2 = vector<T1, N>; \
  typedef v##T2<1> v##T2##1; typedef v##T2<2> v##T2##2; typedef v##T2<3> v##T2##3; typedef v##T2<4> v##T2##4; \
  typedef v##T2<8> v##T2##8; typedef v##T2<16> v##T2##16;
VPP_ALIAS_DECL(char, char) VPP_ALIAS_DECL(short, short) VPP_ALIAS_DECL(int, int) VPP_ALIAS_DECL(float, float) VPP_ALIAS_DECL(double, double)
VPP_ALIAS_DECL(unsigned char, uchar) VPP_ALIAS_DECL(unsigned short, ushort) VPP_ALIAS_DECL(unsigned int, uint)
#undef VPP_ALIAS_DECL

// plus_promotion (vector.hh:36-50): the type of x + x, per component
template <class T> struct plus_promotion_ { typedef decltype(T() + T()) type; };
template <class X, unsigned N> struct plus_promotion_<vector<X, N>> { typedef vector<decltype(X() + X()), N> type; };
template <class T> using plus_promotion = typename plus_promotion_<T>::type;

template <class V> struct cast_to_float_ { typedef float ret; };
template <class X, unsigned N> struct cast_to_float_<vector<X, N>> { typedef vector<float, N> ret; };
template <class V> using cast_to_float = typename cast_to_float_<V>::ret;

template <class V> struct zero { VPP_HD operator V() { return V(0); } };
template <class X, unsigned N> struct zero<vector<X, N>> { VPP_HD operator vector<X, N>() { return vector<X, N>::Zero(); } };

// cast<U>(v) (vector.hh:55-109): scalar<->scalar, vector<->vector of the same size, size-1 vector <-> scalar
namespace detail {
template <class T> struct is_vector : std::false_type {};
template <class X, unsigned N> struct is_vector<vector<X, N>> : std::true_type {};
}  // namespace detail
template <class U, class V> VPP_HD typename std::enable_if<!detail::is_vector<U>::value && !detail::is_vector<V>::value, U>::type cast(const V& v) { return U(v); }
template <class U, class X, unsigned N> VPP_HD typename std::enable_if<detail::is_vector<U>::value, U>::type cast(const vector<X, N>& v) { return v.template cast<typename U::Scalar>(); }
template <class U, class X> VPP_HD typename std::enable_if<!detail::is_vector<U>::value, U>::type cast(const vector<X, 1>& v) { return U(v[0]); }
template <class U, class V> VPP_HD typename std::enable_if<detail::is_vector<U>::value && !detail::is_vector<V>::value, U>::type cast(const V& v) { U r; r[0] = typename U::Scalar(v); return r; }

static_assert(sizeof(vuchar3) == 3 && sizeof(vfloat2) == 8 && sizeof(vint1) == 4, "pixel vectors are exactly their components: the pitches depend on it");
static_assert(std::is_trivially_copyable<vuchar3>::value && std::is_standard_layout<vuchar3>::value, "pixel vectors travel by memcpy / as kernel arguments");

// pixel-type traits used by the device glue: component type + channel count
template <class V> struct pixel_traits { typedef V component; enum { channels = 1 }; };
template <class X, unsigned N> struct pixel_traits<vector<X, N>> { typedef X component; enum { channels = N }; };

}  // namespace vpp

namespace Eigen {
namespace vpp_detail {
template <class T, int R, int C> struct column_vector;   // only column vectors exist on this path (vector.hh:10: Matrix<T, N, 1>)
template <class T, int R> struct column_vector<T, R, 1> { typedef vpp::vector<T, unsigned(R)> type; };
}  // namespace vpp_detail
template <class T, int R, int C = 1> using Matrix = typename vpp_detail::column_vector<T, R, C>::type;
}  // namespace Eigen
