import torch


class LightningModule(torch.nn.Module):
    pass
