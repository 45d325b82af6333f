# generic bf16 dW kernel (stage 0: N, K in {128, 256, 384}): blocks per round x rounds (LAB library)
export NT_LAB_DTYPE=bf16 TN_LAB_STAGES=0 TN_LAB_LIB=lab
for cfg in "512 2" "512 1" "768 1" "768 2" "1024 1"; do set -- $cfg; echo "== slots $1 rounds $2"; HWGAT_TNB_SLOTS=$1 HWGAT_TNB_ROUNDS=$2 python tools/tn_lab.py 2>&1 | grep -v amdgpu | cut -c1-120; done
