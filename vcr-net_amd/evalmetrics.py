"""Eval-harness arithmetic of the reference: the per-batch accumulation of
``test_one_epoch`` (model/vcrnet_model.py:546-649) and the final figures / log line of
``testVCRNet`` (:768-799).  Host-side bookkeeping over the poses the HIP path produces; with
process-per-GPU sharding the per-rank sums are combined with one all-reduce (``merge``)."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List

import numpy as np
import torch
from scipy.spatial.transform import Rotation


def npmat2euler(mats: np.ndarray, seq: str = "zyx") -> np.ndarray:
    """util/util.py:99-104 (``Rotation.from_dcm`` was renamed ``from_matrix`` in SciPy >= 1.4)."""
    return np.asarray([Rotation.from_matrix(m).as_euler(seq, degrees=True) for m in mats], dtype="float32")


def transform_point_cloud(p: torch.Tensor, R: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    return torch.matmul(R, p) + t.unsqueeze(2)                              # util/util.py:91-96


@dataclass
class EvalAccumulator:
    """Running sums of test_one_epoch.  ``loss`` is args.loss: 'pose' (util/initPara.py default), 'point', anything
    else = pose + 0.1 * point loss on the full clouds (vcrnet_model.py:597-605)."""
    cycle: bool = False
    loss: str = "pose"
    num_examples: int = 0
    sums: Dict[str, float] = field(default_factory=lambda: dict(loss=0.0, loss_vcr=0.0, cycle=0.0, mse_ab=0.0,
                                                                 mae_ab=0.0, mse_ba=0.0, mae_ba=0.0))
    R_gt: List[np.ndarray] = field(default_factory=list)
    t_gt: List[np.ndarray] = field(default_factory=list)
    R_pred: List[np.ndarray] = field(default_factory=list)
    t_pred: List[np.ndarray] = field(default_factory=list)
    euler_gt: List[np.ndarray] = field(default_factory=list)
    Rba_pred: List[np.ndarray] = field(default_factory=list)
    tba_pred: List[np.ndarray] = field(default_factory=list)

    def add_batch(self, src, tgt, R_ab, t_ab, euler_ab, out) -> None:
        """``out`` = (srcK, src_corrK, R_ab_pred, t_ab_pred, R_ba_pred, t_ba_pred) of vcrnetIter."""
        srcK, corrK, Rp, tp, Rbp, tbp = out
        B = src.shape[0]
        self.num_examples += B
        eye = torch.eye(3, device=Rp.device).unsqueeze(0).repeat(B, 1, 1)
        mse = torch.nn.functional.mse_loss
        loss_pose = mse(torch.matmul(Rp.transpose(2, 1), R_ab), eye) + mse(tp, t_ab)      # :606-607
        tsrcK = transform_point_cloud(srcK, R_ab, t_ab)                                     # :585
        if self.loss == "pose":                                                             # :597-605
            loss_vcr = loss_pose
        elif self.loss == "point":
            loss_vcr = mse(tsrcK, corrK)
        else:
            loss_vcr = loss_pose + 0.1 * mse(transform_point_cloud(src, Rp, tp), tgt)
        self.sums["loss_vcr"] += loss_vcr.item() * B                                        # :609
        if self.cycle:                                                                      # :611-620
            rot = mse(torch.matmul(Rbp, Rp), eye)
            tr = torch.mean((torch.matmul(Rbp.transpose(2, 1), tp.view(B, 3, 1)).view(B, 3) + tbp) ** 2, dim=[0, 1])
            cyc = rot + tr
            loss_pose = loss_pose + cyc * 0.1
            self.sums["cycle"] += cyc.item() * 0.1 * B
        self.sums["loss"] += loss_pose.item() * B
        ttgt = transform_point_cloud(tgt, Rbp, tbp)                                         # :583
        self.sums["mse_ab"] += torch.mean((tsrcK - corrK) ** 2).item() * B                 # :626-627
        self.sums["mae_ab"] += torch.mean(torch.abs(tsrcK - corrK)).item() * B
        self.sums["mse_ba"] += torch.mean((ttgt - src) ** 2).item() * B                    # :629-630
        self.sums["mae_ba"] += torch.mean(torch.abs(ttgt - src)).item() * B
        self.R_gt.append(R_ab.detach().cpu().numpy()); self.t_gt.append(t_ab.detach().cpu().numpy())
        self.R_pred.append(Rp.detach().cpu().numpy()); self.t_pred.append(tp.detach().cpu().numpy())
        self.euler_gt.append(np.asarray(euler_ab.cpu() if torch.is_tensor(euler_ab) else euler_ab))
        self.Rba_pred.append(Rbp.detach().cpu().numpy()); self.tba_pred.append(tbp.detach().cpu().numpy())   # :577-580

    def returns(self):
        """The 17-tuple test_one_epoch returns (vcrnet_model.py:645-649), in its order: mean pose loss (with the cycle
        term when args.cycle), mean cycle loss, mse_ab, mae_ab, mse_ba, mae_ba, the stacked labels / predictions of
        both directions, the Euler labels, mean args.loss."""
        n = self.num_examples
        Rg, tg, eg = np.concatenate(self.R_gt, 0), np.concatenate(self.t_gt, 0), np.concatenate(self.euler_gt, 0)
        R_ba = np.swapaxes(Rg, 1, 2)
        t_ba = -np.einsum("bij,bj->bi", R_ba, tg)
        s = self.sums
        return (s["loss"] / n, s["cycle"] / n, s["mse_ab"] / n, s["mae_ab"] / n, s["mse_ba"] / n, s["mae_ba"] / n,
                Rg, tg, np.concatenate(self.R_pred, 0), np.concatenate(self.t_pred, 0), R_ba, t_ba,
                np.concatenate(self.Rba_pred, 0), np.concatenate(self.tba_pred, 0), eg, -eg[:, ::-1], s["loss_vcr"] / n)

    def final_ba(self) -> Dict[str, float]:
        """testVCRNet's B -> A figures, :775,781-790: the predicted inverse pose against the loader's B -> A labels
        (R_ba = R_ab^T, t_ba = -R_ba t_ab, euler_ba = -euler_ab[::-1]; util/data.py:278,286,295), Euler angles of the
        prediction taken in 'xyz' order (:781)."""
        n = max(1, self.num_examples)
        Rg, tg, eg = np.concatenate(self.R_gt, 0), np.concatenate(self.t_gt, 0), np.concatenate(self.euler_gt, 0)
        Rb, tb = np.concatenate(self.Rba_pred, 0), np.concatenate(self.tba_pred, 0)
        t_ba = -np.einsum("bji,bj->bi", Rg, tg)                             # -R_ab^T t_ab
        d = npmat2euler(Rb, "xyz") - np.degrees(-eg[:, ::-1])
        r_mse = np.mean(d ** 2)                 # np.float32 scalars, like the reference's: sqrt is taken in fp32
        t_mse = np.mean((t_ba - tb) ** 2)
        mse_ba = self.sums["mse_ba"] / n
        return {"loss": self.sums["loss_vcr"] / n, "loss_pose": self.sums["loss"] / n, "mse": mse_ba,
                "rmse": float(np.sqrt(mse_ba)), "mae": self.sums["mae_ba"] / n, "rot_mse": float(r_mse),
                "rot_rmse": float(np.sqrt(r_mse)), "rot_mae": float(np.mean(np.abs(d))), "trans_mse": float(t_mse),
                "trans_rmse": float(np.sqrt(t_mse)), "trans_mae": float(np.mean(np.abs(t_ba - tb)))}

    def final(self) -> Dict[str, float]:
        """testVCRNet, :769-790 (A -> B direction)."""
        n = max(1, self.num_examples)
        Rp, tp = np.concatenate(self.R_pred, 0), np.concatenate(self.t_pred, 0)
        tg, eg = np.concatenate(self.t_gt, 0), np.concatenate(self.euler_gt, 0)
        e = npmat2euler(Rp)
        d = e - np.degrees(eg)
        r_mse = np.mean(d ** 2)                 # np.float32 scalars, like the reference's: sqrt is taken in fp32
        t_mse = np.mean((tg - tp) ** 2)
        mse_ab = self.sums["mse_ab"] / n
        return {"loss": self.sums["loss_vcr"] / n, "loss_pose": self.sums["loss"] / n,
                "cycle_loss": self.sums["cycle"] / n, "mse": mse_ab, "rmse": float(np.sqrt(mse_ab)),
                "mae": self.sums["mae_ab"] / n, "rot_mse": float(r_mse), "rot_rmse": float(np.sqrt(r_mse)),
                "rot_mae": float(np.mean(np.abs(d))), "trans_mse": float(t_mse), "trans_rmse": float(np.sqrt(t_mse)),
                "trans_mae": float(np.mean(np.abs(tg - tp)))}

    def merge(self, world: int, device="cpu") -> "EvalAccumulator":
        """Combine the per-rank accumulators of a process-per-GPU run (the replacement for nn.DataParallel's
        gather, util/initPara.py:260): one all-reduce of the running sums, one all-gather of the
        [n,12+12+3+12] per-pair records (ragged shards padded, rank order kept).  Every rank returns the same
        merged accumulator; world == 1 returns self."""
        if world == 1:
            return self
        import torch.distributed as dist

        from . import shard
        keys = sorted(self.sums)
        vec = torch.tensor([self.sums[k] for k in keys] + [float(self.num_examples)], dtype=torch.float64, device=device)
        dist.all_reduce(vec)
        counts = torch.zeros(world, dtype=torch.int64, device=device)
        counts[dist.get_rank()] = self.num_examples
        dist.all_reduce(counts)
        cat = lambda xs, w: (np.concatenate(xs, 0).reshape(-1, w) if xs else np.zeros((0, w), np.float32))
        rec = np.concatenate((cat(self.R_gt, 9), cat(self.t_gt, 3), cat(self.R_pred, 9), cat(self.t_pred, 3),
                              cat(self.euler_gt, 3), cat(self.Rba_pred, 9), cat(self.tba_pred, 3)), 1).astype(np.float32)
        per = int(counts.max().item())
        pad = torch.zeros((per, rec.shape[1]), dtype=torch.float32, device=device)
        pad[: rec.shape[0]] = torch.from_numpy(rec).to(device)
        g = shard.all_gather_poses(pad, world).view(world, per, -1).cpu().numpy()
        allrec = np.concatenate([g[r, : int(counts[r].item())] for r in range(world)], 0)
        out = EvalAccumulator(cycle=self.cycle, loss=self.loss, num_examples=int(round(vec[-1].item())))
        out.sums = {k: float(vec[i].item()) for i, k in enumerate(keys)}
        out.R_gt, out.t_gt = [allrec[:, 0:9].reshape(-1, 3, 3)], [allrec[:, 9:12]]
        out.R_pred, out.t_pred = [allrec[:, 12:21].reshape(-1, 3, 3)], [allrec[:, 21:24]]
        out.euler_gt = [allrec[:, 24:27]]
        out.Rba_pred, out.tba_pred = [allrec[:, 27:36].reshape(-1, 3, 3)], [allrec[:, 36:39]]
        return out

    @staticmethod
    def format_final(m: Dict[str, float], epoch: int = -1) -> str:
        """The '==FINAL TEST== / A--------->B' line of :792-799."""
        return ("EPOCH:: %d, Loss: %f, test_LossPose: %f, Cycle Loss: %f, MSE: %f, RMSE: %f, MAE: %f, rot_MSE: %f, "
                "rot_RMSE: %f, rot_MAE: %f, trans_MSE: %f, trans_RMSE: %f, trans_MAE: %f"
                % (epoch, m["loss"], m["loss_pose"], m["cycle_loss"], m["mse"], m["rmse"], m["mae"], m["rot_mse"],
                   m["rot_rmse"], m["rot_mae"], m["trans_mse"], m["trans_rmse"], m["trans_mae"]))

    @staticmethod
    def format_final_ba(m: Dict[str, float], epoch: int = -1) -> str:
        """The 'B--------->A' line of :800-806, printed when args.cycle.  (The reference's format string there has one
        conversion fewer than arguments -- 'test_LossPose' lost its ': %f' -- so the reference itself raises TypeError
        on this line; the twelve values it passes are printed here with the missing conversion restored.)"""
        return ("EPOCH:: %d, Loss: %f, test_LossPose: %f, MSE: %f, RMSE: %f, MAE: %f, rot_MSE: %f, rot_RMSE: %f, "
                "rot_MAE: %f, trans_MSE: %f, trans_RMSE: %f, trans_MAE: %f"
                % (epoch, m["loss"], m["loss_pose"], m["mse"], m["rmse"], m["mae"], m["rot_mse"], m["rot_rmse"],
                   m["rot_mae"], m["trans_mse"], m["trans_rmse"], m["trans_mae"]))
