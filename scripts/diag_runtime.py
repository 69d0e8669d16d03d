import sys, os
sys.path.insert(0, os.getcwd())
order = sys.argv[1]
def maps():
    return sorted({l.split()[-1] for l in open('/proc/self/maps') if 'amdhip64' in l or 'hsa-runtime' in l})
if order == 'torch_first':
    import torch; torch.cuda.init(); print('torch ok', torch.cuda.device_count())
    from cuda_audio_amd.engine import Convolution
    c = Convolution('d', 4096, max_batch=4); print('engine ok')
    s = torch.cuda.Stream(); print('stream ptr', s.cuda_stream)
    c.set_stream(s.cuda_stream); print('set_stream ok')
else:
    from cuda_audio_amd.engine import Convolution
    c = Convolution('d', 4096, max_batch=4); print('engine ok')
    import torch
    try:
        torch.cuda.init(); print('torch ok')
    except Exception as e: print('torch fail', e)
print(maps())
