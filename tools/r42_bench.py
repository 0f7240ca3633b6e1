import torch, competitive_rl_amd as crl, os
for K in (1, 4):
    env = crl.HipPongVecEnv(65536, seed=0, mode="wrapped", resized_dim=42, frame_stack=K)
    env.reset()
    acts = [torch.randint(0, 3, (65536, 2), device="cuda", dtype=torch.int32) for _ in range(8)]
    for i in range(20): env.step_device(acts[i % 8])
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for i in range(200): env.step_device(acts[i % 8])
    b.record(); torch.cuda.synchronize()
    print("R=42 K=%d: %.1f us/step, %.1f M env-steps/s" % (K, a.elapsed_time(b) / 200 * 1e3, 65536 / (a.elapsed_time(b) / 200) / 1e3))
    env.close()
