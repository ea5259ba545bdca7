// Host-side cold parameter preparation: model pieces -> pi, T, emission table.
#pragma once
#include <stdexcept>
#include <vector>

namespace smcpp_host {

struct ModelParams {
    std::vector<double> a, s;   // piece sizes and lengths (ParameterVector, _smcpp.pyx:66-83)
};

class OnePopPrep {
public:
    OnePopPrep(int n, const std::vector<double> &hs, double polarization_error)
        : n_(n), hs_(hs), pol_(polarization_error) {}
    void compute(const ModelParams &, double, double, double, const std::vector<int> &, int,
                 std::vector<double> &, std::vector<double> &, std::vector<double> &) {
        throw std::runtime_error("host parameter preparation is not built yet: use set_raw");
    }
private:
    int n_;
    std::vector<double> hs_;
    double pol_;
};

}  // namespace smcpp_host
