// detqmcptsdwgpu: the reference's REPLICA-EXCHANGE program (one MPI rank per replica) with every replica running on an MI355X.
//
// BUILT ONLY WHERE /root/reference EXISTS (Makefile target `detqmcptsdwgpu`); test infrastructure for the drop-in boundary.
// The driver is the reference's own DetQMCPT<Model, ModelParams> (src/detqmcpt.h: replica exchange over Boost.MPI, per-parameter
// observable handlers src/mpiobservablehandlerpt.cpp, exchange statistics, configuration streams per control parameter), its
// option parser configureSimulation (src/mpimaindetqmcptsdwopdim.cpp, pulled in with main() renamed and the
// DetQMCPT<DetSDW<...>> instantiations compiled out by the reference's DETSDW_NO_O* switches), vendored Boost.MPI on the MPICH
// of this image.  The model is DetSDWGpu (detsdwgpu.h) over libdetqmc_amd.so.
//     mpiexec -n 4 detqmcptsdwgpu -c simulation.conf
// DQMC_DEVICE selects the GPU of a rank (default 0: all replicas on one card, each in its own process).
#define DETSDW_NO_O1
#define DETSDW_NO_O2
#define DETSDW_NO_O3
#define main reference_detqmcptsdw_main
#include "mpimaindetqmcptsdwopdim.cpp"
#undef main
#include "detsdwgpu.h"

int main(int argc, char** argv) {
    boost::mpi::environment env(argc, argv);
    boost::mpi::communicator world;
    if (world.rank() == 0) std::cout << "Build info:\n" << metadataToString(collectVersionInfo()) << "\n";
    DetModelLoggingParams parlogging;
    ModelParamsDetSDW parmodel;
    DetQMCParams parmc;
    DetQMCPTParams parpt;
    bool runSimulation, resumeSimulation;
    try {
        std::tie(runSimulation, resumeSimulation, parlogging, parmodel, parmc, parpt) = configureSimulation(argc, argv);
        if (!runSimulation) return 0;
        // one GPU per rank when the node has several (rank = device), else all ranks share device 0
        if (!std::getenv("DQMC_DEVICE") && std::getenv("DQMC_DEVICE_PER_RANK"))
            setenv("DQMC_DEVICE", std::to_string(world.rank()).c_str(), 1);
        timing.start("total");
        if (!resumeSimulation) {
            DetQMCPT<DetSDWGpu, ModelParamsDetSDW> simulation(parmodel, parmc, parpt, parlogging);
            simulation.run();
        } else {
            DetQMCPT<DetSDWGpu, ModelParamsDetSDW> simulation(parmc.stateFileName, parmc);
            simulation.run();
        }
        timing.stop("total");
    } catch (const std::exception& e) {
        std::cerr << "detqmcptsdwgpu (rank " << world.rank() << "): " << e.what() << "\n";
        env.abort(1);
    }
    return 0;
}
