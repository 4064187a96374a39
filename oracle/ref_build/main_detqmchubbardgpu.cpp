// detqmchubbardgpu: the reference's Hubbard simulation program (BASELINE config 1) with the replica running on an MI355X.
// BUILT ONLY WHERE /root/reference EXISTS (Makefile target `detqmchubbardgpu`).  The option parser is the reference's
// configureSimulation (src/maindetqmchubbard.cpp, pulled in with main() renamed -- that file instantiates DetQMC<DetHubbard>
// unconditionally, so the reference's CPU model is linked in too and stays unused), the driver its DetQMC<> template, the
// model DetHubbardGpu (dethubbardgpu.h) over libdetqmc_amd.so.
#define main reference_detqmchubbard_main
#include "maindetqmchubbard.cpp"
#undef main
#include "dethubbardgpu.h"

int main(int argc, char** argv) {
    std::cout << "Build info:\n" << metadataToString(collectVersionInfo()) << "\n";
    ModelParams<DetHubbard> parmodel;
    DetQMCParams parmc;
    bool runSimulation, resumeSimulation;
    try {
        std::tie(runSimulation, resumeSimulation, parmodel, parmc) = configureSimulation(argc, argv);
        if (!runSimulation) return 0;
        timing.start("total");
        if (!resumeSimulation) {
            DetQMC<DetHubbardGpu, ModelParams<DetHubbard>> simulation(parmodel, parmc);
            simulation.run();
        } else {
            DetQMC<DetHubbardGpu, ModelParams<DetHubbard>> simulation(parmc.stateFileName, parmc);
            simulation.run();
        }
        timing.stop("total");
    } catch (const std::exception& e) {
        std::cerr << "detqmchubbardgpu: " << e.what() << "\n";
        return 1;
    }
    return 0;
}
