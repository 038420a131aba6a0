import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "oracle")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import casapose_oracle as O, torch_train_ref as R
import test_gpu_train as T
dev = torch.device("cuda:0")
b, h, w, k = 2, 64, 96, 4
part, guid = O.VARIANTS["casapose_c_gcu5"]
params, store, plan, img, lab, kpts = T._setup(dev, b, h, w, k, partial=part, guided=guid, bilinear=(False,) * 5, sharing={})
plan.refresh_weights(torch.cuda.current_stream(dev).cuda_stream)
labd = torch.from_numpy(lab).to(dev)
out = plan.forward(torch.from_numpy(img).to(dev), cond_labels=labd)
p64 = R.to_torch(params)
ref = R.forward_train(p64, torch.from_numpy(img.astype(np.float64)), torch.from_numpy(lab.astype(np.int64)), {}, partial=part, guided=guid, bilinear=(False,) * 5)
got = out.cpu().numpy()
print("out err", T.rel(got[..., :k], ref.detach().numpy()[..., :k]), T.rel(got[..., k:], ref.detach().numpy()[..., k:]))
wts = (1.0, 0.5, 0.015)
plan.loss_and_grad(labd, labd, torch.from_numpy(kpts).to(dev), *wts, filter_with_segmentation=False)
ml, vl, pl = R.losses(ref, torch.from_numpy(lab.astype(np.int64)), torch.from_numpy(kpts.astype(np.float64)), k, 9, False)
(wts[0] * ml + wts[1] * vl + wts[2] * pl).backward()
plan.backward(); torch.cuda.synchronize()
worst = {n: T.rel_l2(store.grad_view(n).cpu().numpy(), p64[n].grad.numpy()) for n in store.offsets}
v = sorted(worst.values())
print("grad relL2: median %.3g  90%% %.3g  max %.3g" % (np.median(v), v[int(0.9 * len(v))], v[-1]))
for n in ["pv_final_conv_vertex.kernel", "pv_block_10_prepare_conv2d.weights", "pv_block_6_prepare_conv2d.weights", "pv_block_5_conv2d.kernel", "pv_block_1_conv2d.kernel", "stage4_unit2_conv2.kernel", "stage1_unit1_conv1.kernel", "conv0.kernel"]:
    print("  %-40s %.3g" % (n, worst.get(n, float("nan"))))
