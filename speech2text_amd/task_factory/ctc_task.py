"""CTC task (reference task_factory/ctc_task.py:32-227): cmvn -> encoder -> decoder -> CTC."""
import torch
import torch.nn.functional as F

from speech2text_amd.model.decoder.decoder import Decoder
from speech2text_amd.model.encoder.encoder import Encoder
from speech2text_amd.model.loss.loss import Loss
from speech2text_amd.task_factory.base import TaskBase


class CtcTask(TaskBase):
    def __init__(self, config) -> None:
        super().__init__(config)
        self._encoder = Encoder(config["encoder"])
        self._decoder = Decoder(config["decoder"])
        self._loss = Loss(config["loss"])
        self._metric = self._asr_metric()

    def _optimizer_params(self):
        """reference ctc_task.py:201-213: encoder / decoder groups under seperate_lr."""
        sep = self._optim_config["seperate_lr"]
        if sep["apply"]:
            c = sep["config"]
            return [{"params": self._encoder.parameters(), "name": "encoder_lr", "lr": c["encoder_lr"]},
                    {"params": self._decoder.parameters(), "name": "decoder_lr", "lr": c["decoder_lr"]}]
        return self.parameters()

    def training_step(self, batch, batch_idx):
        feat, feat_len = self.features(batch)
        enc, enc_len = self._encoder(feat, feat_len)
        dec, dec_len = self._decoder(enc, enc_len)
        loss = self._loss({"logits": dec, "logits_length": dec_len, "targets": batch["label"],
                           "targets_length": batch["label_length"]})
        self.log("train_loss", loss, sync_dist=True, prog_bar=True, logger=True)
        return loss.mean()

    @torch.no_grad()
    def validation_step(self, batch, batch_idx):
        """reference ctc_task.py:159-190: loss + greedy-search WER on log-softmax of the head."""
        feat, feat_len = self.features(batch)
        enc, enc_len = self._encoder(feat, feat_len)
        dec, dec_len = self._decoder(enc, enc_len)
        loss = self._loss({"logits": dec, "logits_length": dec_len, "targets": batch["label"],
                           "targets_length": batch["label_length"]})
        wer = self._wer(F.log_softmax(dec, dim=-1), dec_len, batch["label"])
        self.log_dict({"val_loss": loss, "wer": wer}, sync_dist=True, prog_bar=True)
        return {"val_loss": loss, "wer": wer}
