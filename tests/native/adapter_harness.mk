# tests/native/adapter_harness.mk — TEST INFRASTRUCTURE ONLY.
# Links the product's adapter objects (openmvg_amd/adapter/Makefile -> openmvg_amd/lib/adapter_obj/*.o) with the UNCHANGED
# caller-side shims of oracle/ (ref_shim_{match,ba}.cpp: the same code that drives the reference in oracle/_ref), the
# openMVG objects those callers and adapters still need (taken from oracle/_ref/obj, built by oracle/Makefile from the sources
# under $(REF)) and libmvgx_hip.so. The parity tests then call identical entry points in oracle/_ref/libref_*.so (reference
# TUs) and in tests/native/_build/libmvgx_openmvg_adapter{,_ba}.so (replacement TUs).
# Two libraries, because the two halves are compiled with different Eigen ABIs (AVX2 for the matching objects, none for the
# sfm / geometry objects) and must not share inline Eigen allocation code.
REF ?= /root/reference/src
HERE := $(dir $(abspath $(lastword $(MAKEFILE_LIST))))
ROOT := $(abspath $(HERE)/../..)
OUT := $(ROOT)/tests/native/_build
LIBDIR := $(ROOT)/openmvg_amd/lib
AOBJ := $(LIBDIR)/adapter_obj
REFOBJ := $(ROOT)/oracle/_ref/obj
CXX ?= g++
INC := -I$(ROOT)/include -I$(REF) -I$(REF)/third_party/eigen -I$(REF)/third_party \
       -I$(REF)/third_party/flann/src/cpp -I$(REF)/third_party/hnswlib -I$(REF)/dependencies/cereal/include \
       -I$(ROOT)/openmvg_amd/adapter/ceres_config -I$(REF)/third_party/ceres-solver/include \
       -I$(REF)/third_party/ceres-solver/internal/ceres/miniglog
BASEFLAGS := -std=c++11 -O3 -fPIC -fopenmp -DOPENMVG_USE_OPENMP -DEIGEN_MPL2_ONLY -w $(INC)
REF_MATCH_OBJS := $(REFOBJ)/openMVG/matching/regions_matcher.o $(REFOBJ)/openMVG/features/feature.o \
            $(REFOBJ)/third_party/stlplus3/filesystemSimplified/file_system.o \
            $(REFOBJ)/third_party/stlplus3/filesystemSimplified/portability_fixes.o \
            $(REFOBJ)/third_party/stlplus3/filesystemSimplified/wildcard.o
REF_BA_OBJS := $(REFOBJ)/ba/openMVG/numeric/numeric.o $(REFOBJ)/ba/openMVG/sfm/sfm_data_transform.o \
            $(REFOBJ)/ba/openMVG/geometry/Similarity3.o $(REFOBJ)/ba/openMVG/geometry/Similarity3_Kernel.o \
            $(REFOBJ)/ba/openMVG/geometry/rigid_transformation3D_srt.o \
            $(REFOBJ)/ba/openMVG/sfm/sfm_data_filters.o $(REFOBJ)/ba/openMVG/multiview/projection.o \
            $(REFOBJ)/third_party/stlplus3/filesystemSimplified/file_system.o \
            $(REFOBJ)/third_party/stlplus3/filesystemSimplified/portability_fixes.o \
            $(REFOBJ)/third_party/stlplus3/filesystemSimplified/wildcard.o
# the BA adapter libraries: sfm_data_filters.cpp compiled as INTEGRATION.md prescribes (its two outlier filters renamed to *_cpu: the
# replacement TU mvgx_outlier_filters.cpp defines the original names and falls back to these)
ADAPTER_BA_REF_OBJS := $(filter-out $(REFOBJ)/ba/openMVG/sfm/sfm_data_filters.o,$(REF_BA_OBJS)) $(OUT)/sfm_data_filters_cpu.o
ADAPTER_BA_OBJS := $(AOBJ)/mvgx_bundle_adjustment.o $(AOBJ)/mvgx_bundle_adjustment_ceres.o $(AOBJ)/mvgx_outlier_filters.o
RPATH := -Wl,-rpath,'$$ORIGIN/../../../openmvg_amd/lib'

all: $(OUT)/libmvgx_openmvg_adapter.so $(OUT)/libmvgx_openmvg_adapter_ba.so $(OUT)/libmvgx_openmvg_adapter_geo.so

# The geometric filter: the caller code of oracle/ref_shim_geofilter.cpp compiled with the adapter's header forced in first
# (what a maintainer does with one #include in main_GeometricFilter.cpp), linked with the specialisation's object and the
# reference sources the caller still needs (compiled here: no AVX2, like the adapter object)
GEO_REF_SRCS := $(REF)/openMVG/multiview/solver_fundamental_kernel.cpp $(REF)/openMVG/multiview/solver_homography_kernel.cpp $(REF)/openMVG/numeric/nullspace.cpp \
            $(REF)/openMVG/multiview/solver_essential_five_point.cpp $(REF)/openMVG/multiview/solver_essential_kernel.cpp $(REF)/openMVG/multiview/essential.cpp \
            $(REF)/openMVG/multiview/projection.cpp $(REF)/openMVG/multiview/triangulation.cpp $(REF)/openMVG/multiview/solver_essential_three_point.cpp \
            $(REF)/openMVG/multiview/solver_essential_eight_point.cpp $(REF)/openMVG/multiview/motion_from_essential.cpp \
            $(REF)/openMVG/numeric/numeric.cpp $(REF)/openMVG/multiview/conditioning.cpp \
            $(REF)/openMVG/matching_image_collection/Geometric_Filter_utils.cpp $(REF)/openMVG/features/feature.cpp \
            $(REF)/third_party/stlplus3/filesystemSimplified/file_system.cpp $(REF)/third_party/stlplus3/filesystemSimplified/portability_fixes.cpp \
            $(REF)/third_party/stlplus3/filesystemSimplified/wildcard.cpp
GEO_FLAGS := $(BASEFLAGS) -I$(ROOT)/openmvg_amd/adapter -include $(ROOT)/openmvg_amd/adapter/mvgx_geometric_filter.hpp
$(OUT)/libmvgx_openmvg_adapter_geo.so: $(ROOT)/oracle/ref_shim_geofilter.cpp $(AOBJ)/mvgx_geometric_filter.o $(LIBDIR)/libmvgx_hip.so $(lastword $(MAKEFILE_LIST))
	@mkdir -p $(OUT)
	$(CXX) $(GEO_FLAGS) -shared -Wl,-Bsymbolic $(RPATH) -o $@ $(ROOT)/oracle/ref_shim_geofilter.cpp $(GEO_REF_SRCS) $(AOBJ)/mvgx_geometric_filter.o -L$(LIBDIR) -lmvgx_hip -lpthread
$(OUT)/libmvgx_openmvg_adapter_geo_emu.so: $(ROOT)/oracle/ref_shim_geofilter.cpp $(AOBJ)/mvgx_geometric_filter.o $(OUT)/libmvgx_ba_emu.so $(lastword $(MAKEFILE_LIST))
	@mkdir -p $(OUT)
	$(CXX) $(GEO_FLAGS) -shared -Wl,-Bsymbolic -Wl,-rpath,'$$ORIGIN' -o $@ $(ROOT)/oracle/ref_shim_geofilter.cpp $(GEO_REF_SRCS) $(AOBJ)/mvgx_geometric_filter.o -L$(OUT) -lmvgx_ba_emu -lpthread
emu: $(OUT)/libmvgx_openmvg_adapter_geo_emu.so

$(OUT)/ref_shim_match.o: $(ROOT)/oracle/ref_shim_match.cpp
	@mkdir -p $(OUT)
	$(CXX) $(BASEFLAGS) -DOPENMVG_USE_AVX2 -DOPENMVG_USE_AVX -mavx2 -c $< -o $@
$(OUT)/ref_shim_ba.o: $(ROOT)/oracle/ref_shim_ba.cpp
	@mkdir -p $(OUT)
	$(CXX) $(BASEFLAGS) -c $< -o $@

$(OUT)/sfm_data_filters_cpu.o: $(REF)/openMVG/sfm/sfm_data_filters.cpp
	@mkdir -p $(OUT)
	$(CXX) $(BASEFLAGS) -DRemoveOutliers_PixelResidualError=RemoveOutliers_PixelResidualError_cpu -DRemoveOutliers_AngleError=RemoveOutliers_AngleError_cpu -c $< -o $@

$(OUT)/libmvgx_openmvg_adapter.so: $(OUT)/ref_shim_match.o $(AOBJ)/mvgx_matcher_regions.o $(AOBJ)/mvgx_cascade_hashing_matcher_regions.o $(REF_MATCH_OBJS) $(LIBDIR)/libmvgx_hip.so $(lastword $(MAKEFILE_LIST))
	$(CXX) -shared -fopenmp -Wl,-Bsymbolic $(RPATH) -o $@ $(OUT)/ref_shim_match.o $(AOBJ)/mvgx_matcher_regions.o $(AOBJ)/mvgx_cascade_hashing_matcher_regions.o $(REF_MATCH_OBJS) -L$(LIBDIR) -lmvgx_hip -lpthread

$(OUT)/libmvgx_openmvg_adapter_ba.so: $(OUT)/ref_shim_ba.o $(ADAPTER_BA_OBJS) $(ADAPTER_BA_REF_OBJS) $(LIBDIR)/libmvgx_hip.so $(lastword $(MAKEFILE_LIST))
	$(CXX) -shared -fopenmp -Wl,-Bsymbolic $(RPATH) -o $@ $(OUT)/ref_shim_ba.o $(ADAPTER_BA_OBJS) $(ADAPTER_BA_REF_OBJS) -L$(LIBDIR) -lmvgx_hip -lpthread

# The matcher half once more, linked against the two emulation libraries instead of libmvgx_hip.so: the C++ routing code of the
# adapter (region-type dispatch, batching, delivery threads) runs on the CPU test box (tests/test_adapter_emu_cpu.py).
emu: $(OUT)/libmvgx_openmvg_adapter_emu.so
$(OUT)/libmvgx_openmvg_adapter_emu.so: $(OUT)/ref_shim_match.o $(AOBJ)/mvgx_matcher_regions.o $(AOBJ)/mvgx_cascade_hashing_matcher_regions.o $(REF_MATCH_OBJS) $(OUT)/libmvgx_ba_emu.so $(OUT)/libmvgx_match_emu.so $(lastword $(MAKEFILE_LIST))
	$(CXX) -shared -fopenmp -Wl,-Bsymbolic -Wl,-rpath,'$$ORIGIN' -o $@ $(OUT)/ref_shim_match.o $(AOBJ)/mvgx_matcher_regions.o $(AOBJ)/mvgx_cascade_hashing_matcher_regions.o $(REF_MATCH_OBJS) -L$(OUT) -lmvgx_match_emu -lmvgx_ba_emu -lpthread

$(OUT)/libmvgx_openmvg_adapter_ba_emu.so: $(OUT)/ref_shim_ba.o $(ADAPTER_BA_OBJS) $(ADAPTER_BA_REF_OBJS) $(OUT)/libmvgx_ba_emu.so $(lastword $(MAKEFILE_LIST))
	$(CXX) -shared -fopenmp -Wl,-Bsymbolic -Wl,-rpath,'$$ORIGIN' -o $@ $(OUT)/ref_shim_ba.o $(ADAPTER_BA_OBJS) $(ADAPTER_BA_REF_OBJS) -L$(OUT) -lmvgx_ba_emu -lpthread
emu: $(OUT)/libmvgx_openmvg_adapter_ba_emu.so

clean:
	rm -f $(OUT)/libmvgx_openmvg_adapter.so $(OUT)/libmvgx_openmvg_adapter_ba.so $(OUT)/ref_shim_match.o $(OUT)/ref_shim_ba.o
