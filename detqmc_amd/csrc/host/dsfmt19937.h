// dSFMT-19937 restated from the published algorithm (M. Saito, M. Matsumoto, "A PRNG specialized in
// double precision floating point numbers using an affine transition", MCQMC 2008) with the parameter
// set dSFMT2-19937:117-19:ffafffffffb3f-ffdfffc90fffd, plus the reference's seed mangling and
// (0,1) output map.  The stream must be bit-identical to the reference's RngWrapper
// (src/rngwrapper.h:42-62, src/rngwrapper.cpp:31-49, third-party src/dsfmt/) because the order in
// which uniforms are consumed defines the Markov chain.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace detqmc {

class DSFMT19937 {
public:
    explicit DSFMT19937(uint32_t seed = 0) { init(seed); }
    void init(uint32_t seed);
    // uniform in (0,1): dsfmt_genrand_open_open
    double genrand_open_open();
    // raw generator state (checkpoints)
    static constexpr size_t state_words() { return (N_ + 1) * 2 + 1; }
    void get_state(uint64_t* out) const { for (int i = 0; i < (N_ + 1) * 2; ++i) out[i] = st_[i]; out[(N_ + 1) * 2] = (uint64_t)idx_; }
    void set_state(const uint64_t* in) { for (int i = 0; i < (N_ + 1) * 2; ++i) st_[i] = in[i]; idx_ = (int)in[(N_ + 1) * 2]; }
private:
    static constexpr int N_ = 191;
    uint64_t st_[(N_ + 1) * 2];
    int idx_;
    void gen_rand_all();
};

// RngWrapper semantics + a look-ahead window so a batch of upcoming draws can be shipped to the
// device and only the consumed prefix is retired afterwards.
class RngStream {
public:
    RngStream(uint32_t seed = 0, uint32_t processIndex = 0);
    double rand01();                                       // rngwrapper.h:54-57
    double randRange(double low, double high) { return low + (high - low) * rand01(); }
    int randInt(int low, int high) { return low + static_cast<int>((high - low + 1.0) * rand01()); }   // rngwrapper.h:66-68
    void randPointOnCircle(double& x, double& y);                                                   // rngwrapper.h:82-87
    void randPointOnSphere(double& x, double& y, double& z);                                        // rngwrapper.h:70-80
    const double* peek(size_t n);                          // next n draws, not consumed
    void consume(size_t n);
    uint64_t drawn() const { return drawn_; }
    uint32_t mySeed() const { return mySeed_; }
    // exact stream position for checkpoints (the reference serialises the dSFMT state string, rngwrapper.h:100-116)
    std::vector<uint64_t> serialize() const;
    void deserialize(const std::vector<uint64_t>& blob);
private:
    DSFMT19937 gen_;
    std::vector<double> buf_;
    size_t pos_ = 0;
    uint64_t drawn_ = 0;
    uint32_t mySeed_ = 0;
};

}  // namespace detqmc
