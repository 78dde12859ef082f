export TMPDIR=/tmp
mkdir -p gpurun_out/r05
L=torchaudio-contrib_amd/libtac_amd.so
V=gpurun_variants
timeout 300 env TAC_AB_N=300 python tools/r04/ab_inproc.py stft base=$L newhop=$V/libtac_newhop.so 2>&1 | grep -v "amdgpu.ids\|^check" > gpurun_out/r05/batch9_ab_newhop.txt
timeout 300 env TAC_AB_N=300 python tools/r04/ab_inproc.py spec base=$L newhop=$V/libtac_newhop.so 2>&1 | grep -v "amdgpu.ids\|^check" >> gpurun_out/r05/batch9_ab_newhop.txt
cat gpurun_out/r05/batch9_ab_newhop.txt
