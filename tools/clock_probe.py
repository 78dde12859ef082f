import os, sys, time, torch
sys.path.insert(0, '/root/repo')
import torchaudio_contrib_amd as tac
x = torch.rand(256, 1, 160000, device='cuda') * 2 - 1
mel = torch.nn.Sequential(*tac.Melspectrogram(num_mels=128, sample_rate=16000, fft_length=2048, hop_length=512), tac.AmplitudeToDb()).cuda()
for _ in range(5): mel(x)
torch.cuda.synchronize()
def burst(n):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); mel(x); b.record()
    torch.cuda.synchronize()
    return [a.elapsed_time(b) for a, b in ev]
for idle in (2.0, 0.5, 0.1, 0.0):
    time.sleep(idle)
    ts = burst(2000)
    print('idle %.1fs: first 5 %s | launches 100-110 mean %.4f | 1000-1100 mean %.4f | last 100 mean %.4f' % (
        idle, ' '.join('%.4f' % t for t in ts[:5]), sum(ts[100:110]) / 10, sum(ts[1000:1100]) / 100, sum(ts[-100:]) / 100))
os.system('rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|mclk" | head -6')
