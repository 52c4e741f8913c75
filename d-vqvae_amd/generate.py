"""Generation plumbing shared by the four ``gen_diverse_grasp_*`` entry points (reference:
gen_diverse_grasp_obman.py:194-365 and the ho3d/grab/FHAB variants): model build, checkpoint loading, per-object
random rotations, ONE batched ``GenNet.gen`` call per object (the reference loops B=1 calls), 61-parameter
assembly, the final posed-MANO pass and the per-object JSON the downstream tools read.

Out of scope: the physics / mesh metrics (pybullet, igl, trimesh, V-HACD) and the dataset
readers for /data/ObMan, /data/GRAB_unzip, HO3D models -- objects come from ``--objects *.npy`` point clouds
([N,3], e.g. models/Object_models/*/…resampled.npy) or from the synthetic generator."""
from __future__ import annotations

import argparse
import json
import os
import time
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import dist, mano as dmano, ops, synth
from .network.gen_net import GenNet

CANONICAL_OFFSET = (-0.0793, 0.0208, -0.6924)       # gen_diverse_grasp_ho3d.py:221

DATASETS = {                                         # grasps per object, random rotation per grasp
    "obman": dict(num_grasp=1, rotate=False),        # gen_diverse_grasp_obman.py:233
    "ho3d": dict(num_grasp=100, rotate=True),        # gen_diverse_grasp_ho3d.py:212
    "grab": dict(num_grasp=20, rotate=True),         # gen_diverse_grasp_grab.py:202
    "FHAB": dict(num_grasp=49, rotate=True),         # gen_diverse_grasp_FHAB.py:200
}


def build_parser(dataset: str) -> argparse.ArgumentParser:
    d = DATASETS[dataset]
    p = argparse.ArgumentParser(description=f"batched grasp generation ({dataset})")
    # the reference's ten flags (gen_diverse_grasp_obman.py:310-322); only use_cuda / num_grasp are ever read there
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--batch_size", type=int, default=1)
    p.add_argument("--use_cuda", type=int, default=1)
    p.add_argument("--dataloader_workers", type=int, default=32)
    p.add_argument("--encoder_layer_sizes", type=list, default=[1024, 512, 256])
    p.add_argument("--decoder_layer_sizes", type=list, default=[1024, 256, 61])
    p.add_argument("--latent_size", type=int, default=64)
    p.add_argument("--obj_inchannel", type=int, default=4)
    p.add_argument("--condition_size", type=int, default=1024)
    p.add_argument("--num_grasp", type=int, default=d["num_grasp"])
    # additions
    p.add_argument("--objects", nargs="*", default=[], help="[N,3] point clouds (.npy); synthetic objects if omitted")
    p.add_argument("--num_objects", type=int, default=2, help="number of synthetic objects")
    p.add_argument("--points", type=int, default=3000)
    p.add_argument("--n_embeddings", type=int, default=128, help="codebook rows (128 = reference checkpoints)")
    p.add_argument("--checkpoint", default="./checkpoints/model_best.pth")
    p.add_argument("--prior_checkpoint", default="./checkpoints/LATENT_BLOCK_pixelcnn.pt")
    p.add_argument("--mano_model", default="./models/mano/MANO_RIGHT.pkl")
    p.add_argument("--out_dir", default=f"./diverse_grasp/{dataset}")
    p.add_argument("--device", default=None)
    return p


def rotation_xyz(angles: np.ndarray) -> np.ndarray:
    """Rx @ Ry @ Rz for angles [G,3] (gen_diverse_grasp_ho3d.py:214-219)."""
    cx, sx = np.cos(angles[:, 0]), np.sin(angles[:, 0])
    cy, sy = np.cos(angles[:, 1]), np.sin(angles[:, 1])
    cz, sz = np.cos(angles[:, 2]), np.sin(angles[:, 2])
    one, zero = np.ones_like(cx), np.zeros_like(cx)
    Rx = np.stack([one, zero, zero, zero, cx, -sx, zero, sx, cx], 1).reshape(-1, 3, 3)
    Ry = np.stack([cy, zero, sy, zero, one, zero, -sy, zero, cy], 1).reshape(-1, 3, 3)
    Rz = np.stack([cz, -sz, zero, sz, cz, zero, zero, zero, one], 1).reshape(-1, 3, 3)
    return Rx @ Ry @ Rz


def object_tensor(points_xyz: np.ndarray) -> torch.Tensor:
    """[N,3] cloud -> the datasets' [4,N] tensor: xyz + constant bbox-diagonal channel (dataset/dataset_FHAB.py:50-54)."""
    diag = float(np.linalg.norm(points_xyz.max(0) - points_xyz.min(0)))
    return torch.from_numpy(np.concatenate([points_xyz.T, np.full((1, points_xyz.shape[0]), diag)], 0).astype(np.float32))


def load_model(args, device) -> GenNet:
    net = GenNet(n_embeddings=args.n_embeddings)
    have = os.path.exists(args.checkpoint) and os.path.exists(args.prior_checkpoint)
    if have:                                         # gen_diverse_grasp_obman.py:333-346
        ck = torch.load(args.checkpoint, map_location="cpu")["network"]
        sd = net.state_dict()
        sd.update({k: v for k, v in ck.items() if k in sd})
        net.load_state_dict(sd)
        net.GatedPixelCNN.load_state_dict(torch.load(args.prior_checkpoint, map_location="cpu"))
    else:
        print(f"[generate] checkpoints not found ({args.checkpoint}); using deterministic synthetic weights")
        sd = synth.synthetic_state_dict(net.state_dict(), 1234)
        if args.n_embeddings < sd["GatedPixelCNN.output_conv.2.bias"].numel():
            sd["GatedPixelCNN.output_conv.2.bias"][args.n_embeddings:] = -1e4     # keep codes inside the codebooks
        net.load_state_dict(sd)
    net.eval().to(device)
    if os.path.exists(args.mano_model):
        layer = dmano.load(model_path=args.mano_model, model_type="mano", use_pca=True, num_pca_comps=45,
                           flat_hand_mean=True)
    else:
        print(f"[generate] {args.mano_model} not found; using the synthetic MANO-shaped model")
        layer = dmano.ManoLayer(dmano.synthetic_mano_arrays())
    net.set_rh_mano(layer.to(device))
    return net


@torch.no_grad()
def generate_for_object(net: GenNet, obj4n: torch.Tensor, num_grasp: int, rotate: bool, rng: np.random.Generator,
                        noise: Optional[torch.Tensor] = None, proxies: bool = False, seed: Optional[int] = None,
                        object_index: Optional[int] = None, row0: int = 0) -> Dict[str, object]:
    """num_grasp grasps for one object in ONE batched call.  Returns the reference's JSON fields plus tensors.
    ``seed`` / ``object_index`` / ``row0`` key the prior's device noise (seed, stream = object, global grasp row), so the
    grasps of an object do not depend on which rank generates it or on how its grasps are split into calls.
    ``proxies``: also the per-grasp penetration / contact proxies (contact.grasp_proxies) of the posed hands against
    the (rotated) object clouds -- the cheap on-device stand-in for the scripts' trimesh / pybullet metrics."""
    dev = next(net.parameters()).device
    G = num_grasp
    if rotate:
        angles = rng.random((G, 3)) * np.pi * 2                                    # ho3d.py:214
        R = rotation_xyz(angles)
        t = np.asarray(CANONICAL_OFFSET)
    else:
        angles = np.zeros((G, 3))
        R = np.tile(np.eye(3), (G, 1, 1))
        t = np.zeros(3)
    batch = ops.transform_cloud(obj4n.to(dev).contiguous(), torch.as_tensor(R, dtype=torch.float32, device=dev),
                                torch.as_tensor(t, dtype=torch.float32, device=dev))
    recon, pos = net.gen(batch, noise=noise, seed=seed, row0=row0, stream_id=object_index)
    params = ops.assemble61(recon, pos)                                            # obman.py:243-247
    final = net.rh_mano(betas=params[:, :10], global_orient=params[:, 10:13], hand_pose=params[:, 13:58],
                        transl=params[:, 58:61])                                   # obman.py:252-253
    Rt = np.concatenate([R, np.broadcast_to(t.reshape(1, 3, 1), (G, 3, 1))], axis=2)
    extra = {}
    if proxies:
        from . import contact
        faces = np.asarray(net.rh_mano.faces)
        if faces.size == 0 or int(faces.max()) == 0:
            raise RuntimeError("proxies: the MANO layer has no face list (synthetic model); load MANO_RIGHT.pkl")
        topo = getattr(net, "_hand_topology", None)
        if topo is None or topo.faces.device != dev:
            topo = contact.HandTopology(faces, final.vertices.shape[1], dev)
            object.__setattr__(net, "_hand_topology", topo)
        extra["proxies"] = contact.grasp_proxies(topo, final.vertices, batch[:, :3].transpose(1, 2))
    return {**extra, "params": params, "vertices": final.vertices,
            "json": {"recon_params": [[p] for p in params.cpu().numpy().tolist()],   # [[61 floats]] per grasp, as the reference
                     "R_list": Rt.tolist(), "trans_list": [t.reshape(3, 1).tolist()] * G, "r_list": angles.tolist()}}


def main(dataset: str, argv: Optional[Sequence[str]] = None) -> List[str]:
    args = build_parser(dataset).parse_args(argv)
    rank, local_rank, world = dist.init()
    if not torch.cuda.is_available():
        raise RuntimeError("the HIP path needs a GPU: there is no CPU fallback (use the reference on CPU)")
    device = torch.device(args.device) if args.device else torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    torch.manual_seed(args.seed)
    net = load_model(args, device)
    if args.objects:
        objs = [(os.path.splitext(os.path.basename(p))[0], object_tensor(np.load(p).astype(np.float64))) for p in args.objects]
    else:
        clouds = synth.synthetic_clouds(args.num_objects, args.points, seed=args.seed)
        objs = [(f"synthetic_{i}", clouds[i]) for i in range(args.num_objects)]
    os.makedirs(args.out_dir, exist_ok=True)
    written = []
    lo, hi = dist.shard_range(len(objs), rank, world)                              # objects are independent: shard them
    total_t, total_g = 0.0, 0
    for gi, (name, obj) in enumerate(objs[lo:hi], start=lo):
        torch.cuda.synchronize(device)
        t0 = time.time()
        rng = np.random.default_rng([args.seed, gi])                                # per OBJECT: rotations independent of the sharding
        out = generate_for_object(net, obj, args.num_grasp, DATASETS[dataset]["rotate"], rng, seed=args.seed, object_index=gi)
        torch.cuda.synchronize(device)                                             # the reference times without a sync
        dt = time.time() - t0
        total_t += dt
        total_g += args.num_grasp
        print(f"gen_time: {dt:.4f} s for {args.num_grasp} grasps of {name}")
        path = os.path.join(args.out_dir, f"obj_id_{name}.json")
        with open(path, "w") as f:
            json.dump(out["json"], f)
        written.append(path)
    if total_g:
        print(f"rank {rank}: {total_g} grasps in {total_t:.3f} s ({total_g / max(total_t, 1e-9):.1f} grasps/s incl. first-call packing)")
    dist.barrier()                                       # every rank's files are on disk when any rank returns
    if world > 1:
        dist.shutdown()
    return written
