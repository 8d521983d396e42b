"""cosmo.jl_amd -- MI355X-native ADMM hot path for COSMO (host-side mirror of the reference interface + ctypes
binding of libcosmo_hip.so).  The directory name contains a dot, so import it through the `cosmo_jl_amd` shim at the
repository root (`import cosmo_jl_amd`)."""
from . import _ffi
from . import _chordal
from ._ffi import CosmoHipError, Handle, load_library
from .model import (AbstractConvexCone, AbstractConvexSet, AccuracyActivation, IterActivation, AndersonAccelerator, Type1, Type2, QRDecomp, NormalEquations, RestartedMemory, RollingMemory, NoRegularizer, CliqueGraphMerge, NoMerge, ParentChildMerge, ComplexPsdConeTriangle, EmptyAccelerator, Box, DualExponentialCone, DualPowerCone, ExponentialCone, PowerCone, CGIndirectKKTSolver, CGSingleReductionKKTSolver, CGJacobiKKTSolver, Constraint, IndirectReducedKKTSolverMINRES, MINRESIndirectKKTSolver,
                    Model, Nonnegatives, PsdCone, PsdConeTriangle, QdldlKKTSolver, Result, SecondOrderCone, Settings,
                    ZeroSet, assemble, convex_sets_from_dict, set_csc, balance_cones, cone_costs, optimize, partition_cones_contiguous, optimize_batch, shard_range, update, warm_start_dual, warm_start_primal, warm_start_slack,
                    with_options)
from . import problems

__all__ = ["Handle", "CosmoHipError", "load_library", "Model", "Settings", "Constraint", "ZeroSet", "Nonnegatives", "Box",
           "SecondOrderCone", "PsdCone", "PsdConeTriangle", "AbstractConvexCone", "AbstractConvexSet", "AccuracyActivation", "IterActivation", "AndersonAccelerator", "Type1", "Type2", "QRDecomp", "NormalEquations", "RestartedMemory", "RollingMemory", "NoRegularizer", "EmptyAccelerator", "CliqueGraphMerge", "NoMerge", "ParentChildMerge", "_chordal", "ComplexPsdConeTriangle", "ExponentialCone", "DualExponentialCone", "PowerCone", "DualPowerCone", "assemble", "convex_sets_from_dict", "set_csc", "optimize", "optimize_batch", "shard_range", "balance_cones", "update", "warm_start_primal",
           "warm_start_slack", "warm_start_dual", "with_options", "CGIndirectKKTSolver", "CGSingleReductionKKTSolver", "CGJacobiKKTSolver", "MINRESIndirectKKTSolver",
           "IndirectReducedKKTSolverMINRES", "QdldlKKTSolver", "Result", "problems", "_ffi"]
