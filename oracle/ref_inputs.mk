# oracle/_ref: what the REFERENCE'S OWN asset loaders return for the sample's assets (TEST INFRASTRUCTURE; see ref_inputs_dump.cpp).
# Compiles /root/reference/src/sample/contrib/{stb_image.h,tiny_obj_loader.h} unmodified, from where they lie, into a small dumper and runs it on
# /root/reference/src/sample/res.  Outputs only into oracle/_ref/ (git-ignored).  Build-container only: the GPU box has no /root/reference and
# uses tests/golden/ref_inputs.json.        make -C oracle -f ref_inputs.mk
REF ?= /root/reference
CXX ?= g++
ASSETS = clouds.png tiles_dif.png tiles_nrm.png tiles_spc.png grass_nrm.png grass_spc.png grass_dif.png sky.png sphere.obj

_ref/index.json: _ref/ref_inputs_dump
	./_ref/ref_inputs_dump $(REF)/src/sample/res _ref $(ASSETS) > /dev/null

_ref/ref_inputs_dump: ref_inputs_dump.cpp $(REF)/src/sample/contrib/stb_image.h $(REF)/src/sample/contrib/tiny_obj_loader.h
	mkdir -p _ref
	$(CXX) -O1 -w -I$(REF)/src/sample/contrib -o $@ ref_inputs_dump.cpp

clean:
	rm -rf _ref
