import sys, torch, pytest
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
xs = [torch.full((1 << 28,), float(sys.argv[1]), device='cuda') for _ in range(24)]   # 24 GiB of the pattern
torch.cuda.synchronize()
del xs
sys.exit(pytest.main(['/root/repo/tests/test_gpu_wholeframe.py', '-q', '-m', 'gpu', '-k', 'volsdf and mlp', '-x']))
