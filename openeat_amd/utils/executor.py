"""Epoch loops with the reference's semantics (/root/reference/openeat/utils/executor.py):
loss / accum_grad, clip_grad_norm_(all params, grad_clip), skip the optimiser step on a
non-finite norm, scheduler.step every update, running averages exclude non-finite losses."""
from contextlib import nullcontext

import torch
from torch.nn.utils import clip_grad_norm_

from openeat_amd.optim import FusedAdam
from openeat_amd.utils.common import map_to_device


class Executor:
    def __init__(self):
        self.step = 0

    def train(self, logger, model, optimizer, scheduler, data_loader, device, args, local_rank=0):
        model.train()
        log_interval = args.get("log_interval", 10)
        clip = args.get("grad_clip", 5.0)
        accum_grad = args.get("accum_grad", 1)
        reducer = args.get("grad_reducer", None)          # openeat_amd.ddp.GradAllReduce for data parallel runs
        fused = isinstance(optimizer, FusedAdam)
        if fused:
            optimizer.max_grad_norm = clip
        logger.info("using accumulate grad, new batch size is {} times larger than before".format(accum_grad))
        num_seen_utts, total_loss, total_acc = 0, 0.0, 0.0
        n_batches = len(data_loader)
        multi = reducer is not None and reducer.world > 1
        it = iter(data_loader)
        batch_idx = -1
        while True:
            # The reference wraps the loop in DistributedDataParallel.join() (executor.py:24-29) so that ranks may see
            # different numbers of batches, and skips empty batches per rank (:37-38).  Here the gradient exchange is an
            # explicit collective, so the ranks first agree on what this iteration is: 2 = everyone has a batch, 1 = some
            # rank's batch came back empty (all skip it: equal collective counts, same accumulation phase everywhere),
            # 0 = some rank's loader is exhausted (all stop: the surplus batches of the others are dropped).
            try:
                keys, batch = next(it)
                state = 2 if len(keys) > 0 else 1
            except StopIteration:
                keys, batch, state = (), None, 0
            if multi:
                state = reducer.agree_min(state)
            if state == 0:
                break
            batch_idx += 1
            if state == 1:
                continue
            batch = map_to_device(batch, device)
            num_utts = len(keys)
            boundary = batch_idx % accum_grad == 0
            if reducer is not None:
                reducer.overlap_enabled = boundary           # model.no_sync on the micro-steps in between (executor.py:42-45)
            loss, acc = model(**batch)
            loss = torch.mean(loss) / accum_grad
            acc = None if acc is None else torch.mean(acc)
            if torch.isfinite(loss):
                num_seen_utts += num_utts
                total_loss += loss.item() * accum_grad * num_utts
                if acc is not None:
                    total_acc += acc.item() * num_utts
            loss.backward()
            if boundary:
                if reducer is not None:
                    reducer()
                if fused:
                    optimizer.step()                       # norm, clip, finite check and Adam on device
                else:
                    grad_norm = clip_grad_norm_(model.parameters(), clip)
                    if torch.isfinite(grad_norm):
                        optimizer.step()
                optimizer.zero_grad()
                scheduler.step()
                self.step += 1
            if batch_idx % log_interval == 0:
                lr = optimizer.param_groups[0]["lr"]
                msg = "TRAIN Batch[{}/{}] Loss:{:.4f} ALoss:{:.4f} ".format(
                    batch_idx, n_batches, loss.item() * accum_grad, total_loss / max(num_seen_utts, 1))
                if acc is not None:
                    msg += "Acc:{:.4f} AAcc:{:.4f} ".format(acc.item(), total_acc / max(num_seen_utts, 1))
                logger.info(msg + "lr:{:.8f} rank:{}".format(lr, local_rank))
        return total_loss / max(num_seen_utts, 1), total_acc / max(num_seen_utts, 1)

    def cv(self, logger, model, data_loader, device, args, local_rank=0):
        model.eval()
        log_interval = args.get("log_interval", 10)
        num_seen_utts, total_loss, total_acc = 0, 0.0, 0.0
        n_batches = len(data_loader)
        with torch.no_grad():
            for batch_idx, (keys, batch) in enumerate(data_loader):
                batch = map_to_device(batch, device)
                num_utts = len(keys)
                if num_utts == 0:
                    continue
                loss, acc = model(**batch)
                loss = torch.mean(loss)
                acc = None if acc is None else torch.mean(acc)
                if torch.isfinite(loss):
                    num_seen_utts += num_utts
                    total_loss += loss.item() * num_utts
                    if acc is not None:
                        total_acc += acc.item() * num_utts
                if batch_idx % log_interval == 0:
                    msg = "CV Batch[{}/{}] Loss:{:.4f} ALoss:{:.4f} ".format(batch_idx, n_batches, loss.item(),
                                                                            total_loss / max(num_seen_utts, 1))
                    if acc is not None:
                        msg += "Acc:{:.4f} AAcc:{:.4f} rank:{}".format(acc.item(), total_acc / max(num_seen_utts, 1), local_rank)
                    logger.info(msg)
        return total_loss / max(num_seen_utts, 1), total_acc / max(num_seen_utts, 1)
