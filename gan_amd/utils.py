"""Loss-dict factories and the train/val loss plot of the reference (utils.py:32-74): same key strings and
file naming so metrics JSON / figures are drop-in compatible."""
import os


def pix2pix_losses():
    """utils.py:32-40"""
    return {'Generator Total Loss': [],
            'Generator Loss (Primary)': [],
            'Generator Loss (Secondary)': [],
            'Discriminator Loss': []}


def cyclegan_losses():
    """utils.py:42-53"""
    return {'X->Y Generator Loss': [],
            'Y->X Generator Loss': [],
            'Total Cycle Loss': [],
            'Total X->Y Generator Loss': [],
            'Total Y->X Generator Loss': [],
            'Discriminator X Loss': [],
            'Discriminator Y Loss': []}


def make_fig(train, val, title: str, output_path: str):
    """Two line graphs (training / validation loss by epoch) in one PNG named `<title>.png` (utils.py:55-74)."""
    import matplotlib
    matplotlib.use('Agg')
    import matplotlib.pyplot as plt
    plt.figure(figsize=(10, 8), dpi=80)
    plt.plot(train, alpha=0.7, label='Training')
    plt.plot(val, alpha=0.7, label='Validation')
    plt.xlabel('Epoch')
    plt.ylabel('Loss')
    plt.legend()
    plt.title(f'{title}')
    plt.tight_layout()
    os.makedirs(output_path, exist_ok=True)
    plt.savefig(os.path.join(output_path, f'{title}.png'), dpi=200)
    plt.close()
