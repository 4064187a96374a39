#include "dsfmt19937.h"
#include <cmath>
#include <cstring>

namespace detqmc {

namespace {
const int POS1 = 117, SL1 = 19, SR = 12;
const uint64_t MSK1 = 0x000ffafffffffb3fULL, MSK2 = 0x000ffdfffc90fffdULL;
const uint64_t FIX1 = 0x90014964b32f4329ULL, FIX2 = 0x3b8d12ac548a7c7aULL;
const uint64_t PCV1 = 0x3d84e1ac0dc82880ULL, PCV2 = 0x0000000000000001ULL;
const uint64_t LOW_MASK = 0x000FFFFFFFFFFFFFULL, HIGH_CONST = 0x3FF0000000000000ULL;
}

void DSFMT19937::init(uint32_t seed) {
    // 32-bit linear-congruential fill of the whole state (little-endian word order)
    const int n32 = (N_ + 1) * 4;
    uint32_t ps[(N_ + 1) * 4];
    ps[0] = seed;
    for (int i = 1; i < n32; ++i) ps[i] = 1812433253u * (ps[i - 1] ^ (ps[i - 1] >> 30)) + (uint32_t)i;
    for (int j = 0; j < (N_ + 1) * 2; ++j) st_[j] = (uint64_t)ps[2 * j] | ((uint64_t)ps[2 * j + 1] << 32);
    // force the exponent bits: every state word is a double in [1,2)
    for (int j = 0; j < 2 * N_; ++j) st_[j] = (st_[j] & LOW_MASK) | HIGH_CONST;
    // period certification on the last 128-bit word
    uint64_t t0 = st_[2 * N_] ^ FIX1, t1 = st_[2 * N_ + 1] ^ FIX2;
    uint64_t inner = (t0 & PCV1) ^ (t1 & PCV2);
    for (int i = 32; i > 0; i >>= 1) inner ^= inner >> i;
    if ((inner & 1) != 1) st_[2 * N_ + 1] ^= 1;
    idx_ = 2 * N_;
}

void DSFMT19937::gen_rand_all() {
    uint64_t L0 = st_[2 * N_], L1 = st_[2 * N_ + 1];
    for (int i = 0; i < N_; ++i) {
        int bi = i + POS1;
        if (bi >= N_) bi -= N_;
        const uint64_t a0 = st_[2 * i], a1 = st_[2 * i + 1];
        const uint64_t b0 = st_[2 * bi], b1 = st_[2 * bi + 1];
        const uint64_t nL0 = (a0 << SL1) ^ (L1 >> 32) ^ (L1 << 32) ^ b0;
        const uint64_t nL1 = (a1 << SL1) ^ (L0 >> 32) ^ (L0 << 32) ^ b1;
        L0 = nL0; L1 = nL1;
        st_[2 * i] = (L0 >> SR) ^ (L0 & MSK1) ^ a0;
        st_[2 * i + 1] = (L1 >> SR) ^ (L1 & MSK2) ^ a1;
    }
    st_[2 * N_] = L0; st_[2 * N_ + 1] = L1;
}

double DSFMT19937::genrand_open_open() {
    if (idx_ >= 2 * N_) { gen_rand_all(); idx_ = 0; }
    uint64_t u = st_[idx_++] | 1ULL;
    double d;
    std::memcpy(&d, &u, sizeof(d));
    return d - 1.0;
}

RngStream::RngStream(uint32_t seed, uint32_t processIndex) {
    // uint32 arithmetic exactly as rngwrapper.cpp:43
    mySeed_ = ((seed * 181u) * ((processIndex - 83u) * 359u)) % 104729u;
    gen_.init(mySeed_);
}

double RngStream::rand01() {
    ++drawn_;
    if (pos_ < buf_.size()) {
        double v = buf_[pos_++];
        if (pos_ == buf_.size()) { buf_.clear(); pos_ = 0; }
        return v;
    }
    return gen_.genrand_open_open();
}

const double* RngStream::peek(size_t n) {
    if (pos_ > 0) { buf_.erase(buf_.begin(), buf_.begin() + pos_); pos_ = 0; }
    while (buf_.size() < n) buf_.push_back(gen_.genrand_open_open());
    return buf_.data();
}

void RngStream::consume(size_t n) {
    drawn_ += n;
    pos_ += n;
    if (pos_ >= buf_.size()) { buf_.clear(); pos_ = 0; }
}

std::vector<uint64_t> RngStream::serialize() const {
    std::vector<uint64_t> b(DSFMT19937::state_words() + 3 + (buf_.size() - pos_));
    gen_.get_state(b.data());
    size_t o = DSFMT19937::state_words();
    b[o++] = drawn_; b[o++] = mySeed_; b[o++] = buf_.size() - pos_;
    for (size_t i = pos_; i < buf_.size(); ++i) { uint64_t u; std::memcpy(&u, &buf_[i], 8); b[o++] = u; }
    return b;
}
void RngStream::deserialize(const std::vector<uint64_t>& b) {
    gen_.set_state(b.data());
    size_t o = DSFMT19937::state_words();
    drawn_ = b[o++]; mySeed_ = (uint32_t)b[o++];
    const size_t nbuf = (size_t)b[o++];
    buf_.assign(nbuf, 0.0); pos_ = 0;
    for (size_t i = 0; i < nbuf; ++i) std::memcpy(&buf_[i], &b[o++], 8);
}

void RngStream::randPointOnCircle(double& x, double& y) {
    const double phi = randRange(0., 2. * M_PI);
    x = std::cos(phi);
    y = std::sin(phi);
}
void RngStream::randPointOnSphere(double& x, double& y, double& z) {
    const double phi = randRange(0., 2. * M_PI);
    const double costheta = randRange(-1., 1.0);
    const double sintheta = std::sqrt(1. - costheta * costheta);
    x = std::cos(phi) * sintheta;
    y = std::sin(phi) * sintheta;
    z = costheta;
}

}  // namespace detqmc
