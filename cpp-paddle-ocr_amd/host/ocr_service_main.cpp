// ocr_service for Linux: the reference's service executable (/root/reference/src/ocr_service_main.cpp:60-140)
// over the Unix-socket OCRIPCService.  Same four options; --pipe-name takes a socket path.
#include <csignal>
#include <cstdio>
#include <string>

#include "ocr_ipc_service.h"

static PaddleOCR::OCRIPCService* g_service = nullptr;
static void on_signal(int) {
  if (g_service) g_service->stop();
}
static void usage() {
  printf("Usage: ocr_service [options]\n"
         "  --model-dir <path>    model directory (default: ./models)\n"
         "  --pipe-name <path>    Unix-domain socket path (default: /tmp/ocr_service.sock)\n"
         "  --gpu-workers <num>   GPU workers, worker i on GPU i mod #GPUs (default: 0)\n"
         "  --cpu-workers <num>   accepted for compatibility; this build has no CPU path (default: 1)\n"
         "  --help\n");
}
int main(int argc, char** argv) {
  std::string model_dir = "./models", pipe_name = "/tmp/ocr_service.sock";
  int gpu_workers = 0, cpu_workers = 1;
  for (int i = 1; i < argc; ++i) {
    const std::string arg = argv[i];
    if (arg == "--help" || arg == "-h") { usage(); return 0; }
    else if (arg == "--model-dir" && i + 1 < argc) model_dir = argv[++i];
    else if (arg == "--pipe-name" && i + 1 < argc) pipe_name = argv[++i];
    else if (arg == "--gpu-workers" && i + 1 < argc) gpu_workers = std::stoi(argv[++i]);
    else if (arg == "--cpu-workers" && i + 1 < argc) cpu_workers = std::stoi(argv[++i]);
    else { usage(); return 1; }
  }
  try {
    PaddleOCR::OCRIPCService service(model_dir, pipe_name, gpu_workers, cpu_workers);
    g_service = &service;
    signal(SIGINT, on_signal);
    signal(SIGTERM, on_signal);
    if (!service.start()) { fprintf(stderr, "Failed to start OCR IPC service on %s\n", pipe_name.c_str()); return 1; }
    printf("OCR service listening on %s (gpu workers: %d)\n", pipe_name.c_str(), gpu_workers);
    fflush(stdout);
    service.waitUntilStopped();
    g_service = nullptr;
  } catch (const std::exception& e) {
    fprintf(stderr, "fatal: %s\n", e.what());
    return 1;
  }
  printf("OCR service stopped\n");
  return 0;
}
