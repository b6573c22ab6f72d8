import sys, os
sys.path.insert(0, "/root/repo")
import torch
from distributional_rl_navigation_amd.iqn.fused_act import fused_qvals, act_context
from distributional_rl_navigation_amd.iqn.model import ObsEncoder
dev = "cuda:0"
g = torch.Generator(device=dev); g.manual_seed(3)
obs = torch.randn(64, 26, device=dev, generator=g) * 3; taus = torch.rand(64, 32, device=dev, generator=g)
def run(tag, fn):
    net = ObsEncoder(26, 9, seed=11, device=dev)
    with torch.no_grad():
        net.hidden_layer.weight[:, :192].zero_()     # only the leftover path feeds layer 2
        fn(net)
        ref = net.get_qvals(obs, 1.0, taus=taus)
    q = fused_qvals(net, obs, 1.0, taus=taus)
    print(f"{tag:50s} v32 err {float((q-ref).abs().max()):.3e}  |Q| {float(ref.abs().max()):.3f}")
run("only leftover", lambda n: None)
run("W1[192:]=0 (h1 = relu(b1)*feat, tau-independent)", lambda n: (n.cos_embedding.weight[192:].zero_(), n.cos_embedding.bias[192:].abs_()))
run("W1[192:]=0, b1=1 (h1 = feat)", lambda n: (n.cos_embedding.weight[192:].zero_(), n.cos_embedding.bias[192:].fill_(1.0)))
run("W1[192:, k!=0]=0 (cos k=0 == 1)", lambda n: (n.cos_embedding.weight[192:, 1:].zero_()))
run("W1[192:, k!=1]=0", lambda n: (n.cos_embedding.weight[192:, 0:1].zero_(), n.cos_embedding.weight[192:, 2:].zero_()))
run("W1[192:, k!=2]=0", lambda n: (n.cos_embedding.weight[192:, 0:2].zero_(), n.cos_embedding.weight[192:, 3:].zero_()))
run("W1[192:, k!=3]=0", lambda n: (n.cos_embedding.weight[192:, 0:3].zero_(), n.cos_embedding.weight[192:, 4:].zero_()))
run("only feature 192 in W2", lambda n: n.hidden_layer.weight[:, 193:].zero_())
run("only feature 196 in W2", lambda n: (n.hidden_layer.weight[:, 192:196].zero_(), n.hidden_layer.weight[:, 197:].zero_()))
run("only feature 200 in W2", lambda n: (n.hidden_layer.weight[:, 192:200].zero_(), n.hidden_layer.weight[:, 201:].zero_()))
run("only feature 207 in W2", lambda n: n.hidden_layer.weight[:, 192:207].zero_())
