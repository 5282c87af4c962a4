// ThreadSanitizer harness (CPU build, no HIP): eight threads resolve the plan of one cold model at once, then again after a setter --
// what emgpu_sample_dbn_multi_host/_device do with one thread per device.  tests/test_host.py builds and runs it:
//   g++ -std=c++17 -O1 -g -fsanitize=thread -I em_model_manned_bayes_amd/csrc tools/tsan/plan_of_race.cpp em_model_manned_bayes_amd/csrc/emgpu_model.cpp
#include "emgpu_model.hpp"
#include <thread>
#include <vector>
#include <cstdio>
int main(int argc, char **argv) {
    int32_t z[3] = {1, 2, 3};
    for (int round = 0; round < 4; round++) {
        emgpu::Model *m = emgpu::load_txt(argv[1], z, 3, false);
        for (int phase = 0; phase < 2; phase++) {
            if (phase) m->set_prior(0, 1.0);
            std::vector<std::thread> ts;
            std::vector<const void *> got(8);
            for (int i = 0; i < 8; i++) ts.emplace_back([&, i] { got[i] = emgpu::plan_of(*m).get(); });
            for (auto &t : ts) t.join();
            for (int i = 1; i < 8; i++) if (got[i] != got[0]) { printf("different plans\n"); return 1; }
        }
        delete m;
    }
    printf("ok\n");
    return 0;
}
