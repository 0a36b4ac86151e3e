// Name -> solver registry.  Mirrors src/solver_manager.hpp:16-41 (same class and method
// names); the GPU entry is registered under a compile-time switch the way the reference
// guards "quasi-mcp-cuda" with CUDA_ENABLED (solver_manager.hpp:22-24, src/config.h.in:12).
// Only the solver this repository implements is registered: the OR-Tools-backed CPU
// solvers stay in the reference.
#ifndef QMCP_AMD_SOLVER_MANAGER_HPP
#define QMCP_AMD_SOLVER_MANAGER_HPP

#include <map>
#include <memory>
#include <string>
#include <vector>

#include "qmcp-solver/quasi_mcp_hip_solver.hpp"
#include "qmcp-solver/solver.hpp"

class SolverManager {
   public:
    SolverManager() {
#ifdef HIP_ENABLED
        solvers_map_.emplace("quasi-mcp-hip", std::make_unique<qmcp::QuasiMcpHipSolver>());
#endif
        for (const auto& entry : solvers_map_) algorithms_names_.push_back(entry.first);
    }

    // lets an embedding application (or a test) add further solvers under the same surface
    void add(const std::string& name, std::unique_ptr<qmcp::Solver> solver) {
        if (solvers_map_.emplace(name, std::move(solver)).second) {
            algorithms_names_.clear();
            for (const auto& entry : solvers_map_) algorithms_names_.push_back(entry.first);
        }
    }

    qmcp::Solver& get(const std::string& solver_name) const { return *solvers_map_.at(solver_name); }
    bool contains(const std::string& solver_name) {
        return solvers_map_.find(solver_name) != solvers_map_.end();
    }
    const std::vector<std::string>& get_names() const { return algorithms_names_; }

   private:
    std::map<std::string, std::unique_ptr<qmcp::Solver>> solvers_map_;
    std::vector<std::string> algorithms_names_;
};

#endif
