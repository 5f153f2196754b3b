B=tools/micro/build/mac_mfma_check
mkdir -p gpurun_out
{
$B 3 5 20 40 2 256
$B 4 16 33 64 2 512
$B 2 17 19 32 2 64
$B 16 16 100 70 2 1024
$B 16 16 704 64 3 8192
$B 16 16 704 32 3 8192 0
$B 64 64 59 64 3 8192 0
$B 64 64 12 64 5 8192 0
$B 64 8 12 64 5 8192 0
} 2>&1 | tee gpurun_out/mfma1.log
