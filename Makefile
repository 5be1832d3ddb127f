# Top-level conveniences.  The library itself builds from ark_vrf_amd/csrc/Makefile (what __graft_entry__.build() drives).
#   make lib               libavrf.so for gfx950 (hipcc cross-compiles without a GPU) + the oracle's C restatement
#   make stamp             build/HEAD_STAMP = the commit whose build is about to be profiled (run in the build container: the
#                          GPU box gets a snapshot without .git); refuses a dirty tree unless DIRTY=1
#   make profiles R=r6     on a GPU box (gpurun -- 'make profiles R=r6'): every committed profile summary of the round,
#                          regenerated from the current build and stamped with build/HEAD_STAMP (tools/refresh_profiles.sh)
#   make check-model       the exact-integer model of the unsaturated-limb arithmetic (tools/fpu_model.py) and the emulator run of
#                          the generated asm multipliers (tools/gen_fpu_asm.py --check)
R ?= r6
lib:
	$(MAKE) -C ark_vrf_amd/csrc -j8
	$(MAKE) -C oracle
stamp:
	@mkdir -p build
	@if [ -z "$(DIRTY)" ] && [ -n "$$(git status --porcelain --untracked-files=no)" ]; then echo "tracked files differ from HEAD: commit first (or DIRTY=1)"; exit 1; fi
	@git rev-parse HEAD > build/HEAD_STAMP && echo "build/HEAD_STAMP = $$(cat build/HEAD_STAMP)"
profiles:
	bash tools/refresh_profiles.sh $(R) $(WHAT)
check-model:
	python3 tools/fpu_model.py
	python3 tools/gen_fpu_asm.py --check
.PHONY: lib stamp profiles check-model
