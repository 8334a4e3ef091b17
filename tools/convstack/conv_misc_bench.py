import torch, torch.nn.functional as F
dev="cuda:0"
def timed(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/iters*1e3
for name,(cin,cout,H,k,stride,pad) in {"dec nin 256->128 @256":(256,128,256,1,1,0),"dec nin 512->256 @128":(512,256,128,1,1,0),"enc nin 128->256 @128":(128,256,128,1,1,0),"enc nin 256->512 @64":(256,512,64,1,1,0),"attn proj 512 @32":(512,512,32,1,1,0),
    "down 128 @256->128":(128,128,257,3,2,0),"down 256 @128->64":(256,256,129,3,2,0),"down 512 @64->32":(512,512,65,3,2,0),"dec conv_in 16->512 @32":(16,512,32,3,1,1),"enc conv_out 512->32 @32":(512,32,32,3,1,1),"enc conv_in 3->128 @256":(3,128,256,3,1,1)}.items():
    x=torch.randn(16,cin,H,H,device=dev).contiguous(memory_format=torch.channels_last)
    w=torch.randn(cout,cin,k,k,device=dev).contiguous(memory_format=torch.channels_last)
    t=timed(lambda: F.conv2d(x,w,None,stride,pad))
    Ho=(H+2*pad-k)//stride+1
    fl=2*16*Ho*Ho*cin*cout*k*k
    by=(x.numel()+16*cout*Ho*Ho)*4
    print(f"{name}: {t:.0f} us, {fl/t/1e6:.0f} TFLOP/s, {by/t/1e6:.2f} TB/s (HBM ideal {by/5e6:.0f} us)")
