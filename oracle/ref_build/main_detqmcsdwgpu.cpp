// detqmcsdwgpu: the reference's single-replica simulation program with the replica running on an MI355X.
//
// BUILT ONLY WHERE /root/reference EXISTS (Makefile target `detqmcsdwgpu`); test infrastructure for the drop-in boundary,
// not product code.  Everything around the replica is the reference's own code compiled where it lies: the option parser
// (configureSimulation of src/maindetqmcsdwopdim.cpp, pulled in below with its main() renamed and its
// DetQMC<DetSDW<...>> instantiations compiled out by the reference's own DETSDW_NO_O* switches), the driver template
// DetQMC<Model, ModelParams> (src/detqmc.h), observable handlers, metadata and result files.  The model is
// DetSDWGpu (detsdwgpu.h), which forwards to libdetqmc_amd.so.  So
//     detqmcsdwgpu -c simulation.conf
// reads the reference's configuration files and writes the reference's output tree (results.values, *.series,
// configs-phi.binarystream, simulation.state, info.dat).
#define DETSDW_NO_O1
#define DETSDW_NO_O2
#define DETSDW_NO_O3
#define main reference_detqmcsdw_main
#include "maindetqmcsdwopdim.cpp"
#undef main
#include "detsdwgpu.h"

int main(int argc, char** argv) {
    std::cout << "Build info:\n" << metadataToString(collectVersionInfo()) << "\n";
    DetModelLoggingParams parlogging;
    ModelParamsDetSDW parmodel;
    DetQMCParams parmc;
    bool runSimulation, resumeSimulation;
    try {
        std::tie(runSimulation, resumeSimulation, parlogging, parmodel, parmc) = configureSimulation(argc, argv);
        if (!runSimulation) return 0;
        timing.start("total");
        if (!resumeSimulation) {
            DetQMC<DetSDWGpu, ModelParamsDetSDW> simulation(parmodel, parmc, parlogging);
            simulation.run();
        } else {
            DetQMC<DetSDWGpu, ModelParamsDetSDW> simulation(parmc.stateFileName, parmc);
            simulation.run();
        }
        timing.stop("total");
    } catch (const std::exception& e) {
        std::cerr << "detqmcsdwgpu: " << e.what() << "\n";
        return 1;
    }
    return 0;
}
