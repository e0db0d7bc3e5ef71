// Rendezvous key-value stores: the native piece behind
// init_process_group('tcp://...') / ('file://...') / env:// in the reference
// (/root/reference/3.multiprocessing_distributed.py:102,
//  /root/reference/6.distributed_slurm_main.py:93-101, /root/reference/2.distributed.py:98).
//
// TcpStore : rank 0 runs a poll()-driven server thread; every rank (rank 0 too)
//            talks to it over one TCP connection.  Ops: SET, GET (blocks until the
//            key exists or the timeout expires), ADD (atomic int64 counter), CHECK,
//            DELETE, NUMKEYS, COMPARE_SET.  WAIT is GET without the payload.
// FileStore: the same API over one append-only log file on a shared filesystem,
//            serialised with fcntl() record locks (works across nodes on NFS/Lustre).
//
// Exported as a C ABI (pda_store_*) and bound with ctypes in comm/store.py.
#include <arpa/inet.h>
#include <errno.h>
#include <fcntl.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <poll.h>
#include <stdint.h>
#include <string.h>
#include <sys/socket.h>
#include <sys/stat.h>
#include <sys/types.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#define PDA_EXPORT extern "C" __attribute__((visibility("default")))

namespace {

enum Op : uint8_t { OP_SET = 1, OP_GET = 2, OP_ADD = 3, OP_CHECK = 4, OP_DELETE = 5, OP_NUMKEYS = 6, OP_CAS = 7,
                    OP_APPEND = 8 };  // value = value + bytes, in one step on the server (c10d Store::append)

using Clock = std::chrono::steady_clock;

bool send_all(int fd, const void* buf, size_t n) {
  const char* p = static_cast<const char*>(buf);
  while (n > 0) {
    ssize_t r = ::send(fd, p, n, MSG_NOSIGNAL);
    if (r < 0) {
      if (errno == EINTR) continue;
      return false;
    }
    p += r;
    n -= (size_t)r;
  }
  return true;
}

bool recv_all(int fd, void* buf, size_t n, int timeout_ms = -1) {
  char* p = static_cast<char*>(buf);
  while (n > 0) {
    if (timeout_ms >= 0) {
      pollfd pf{fd, POLLIN, 0};
      int pr = ::poll(&pf, 1, timeout_ms);
      if (pr == 0) return false;
      if (pr < 0) {
        if (errno == EINTR) continue;
        return false;
      }
    }
    ssize_t r = ::recv(fd, p, n, 0);
    if (r == 0) return false;
    if (r < 0) {
      if (errno == EINTR) continue;
      return false;
    }
    p += r;
    n -= (size_t)r;
  }
  return true;
}

void put_u32(std::string& s, uint32_t v) { s.append(reinterpret_cast<const char*>(&v), 4); }
void put_i64(std::string& s, int64_t v) { s.append(reinterpret_cast<const char*>(&v), 8); }
void put_str(std::string& s, const std::string& v) {
  put_u32(s, (uint32_t)v.size());
  s.append(v);
}

// ---------------------------------------------------------------- server ----
struct Pending {  // a GET waiting for its key
  int fd;
  std::string key;
  Clock::time_point deadline;
};

struct Conn {
  int fd;
  std::string in;  // bytes received, not yet parsed
};

class TcpServer {
 public:
  explicit TcpServer(int port) : port_(port) {}
  ~TcpServer() { stop(); }

  bool start() {
    lfd_ = ::socket(AF_INET, SOCK_STREAM, 0);
    if (lfd_ < 0) return false;
    int one = 1;
    ::setsockopt(lfd_, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
    sockaddr_in a{};
    a.sin_family = AF_INET;
    a.sin_addr.s_addr = htonl(INADDR_ANY);
    a.sin_port = htons((uint16_t)port_);
    if (::bind(lfd_, reinterpret_cast<sockaddr*>(&a), sizeof(a)) != 0) { ::close(lfd_); lfd_ = -1; return false; }
    if (::listen(lfd_, 1024) != 0) { ::close(lfd_); lfd_ = -1; return false; }
    socklen_t len = sizeof(a);
    ::getsockname(lfd_, reinterpret_cast<sockaddr*>(&a), &len);
    port_ = ntohs(a.sin_port);
    if (::pipe(wake_) != 0) { ::close(lfd_); lfd_ = -1; return false; }
    running_ = true;
    th_ = std::thread([this] { loop(); });
    return true;
  }
  void stop() {
    if (!running_.exchange(false)) return;
    char c = 1;
    (void)!::write(wake_[1], &c, 1);
    if (th_.joinable()) th_.join();
    for (auto& c2 : conns_) ::close(c2.fd);
    conns_.clear();
    if (lfd_ >= 0) ::close(lfd_);
    ::close(wake_[0]);
    ::close(wake_[1]);
    lfd_ = -1;
  }
  int port() const { return port_; }

 private:
  void reply_value(int fd, bool found, const std::string& v) {
    std::string out;
    out.push_back(found ? 1 : 0);
    put_str(out, v);
    send_all(fd, out.data(), out.size());
  }
  void reply_i64(int fd, int64_t v) {
    std::string out;
    put_i64(out, v);
    send_all(fd, out.data(), out.size());
  }
  void key_arrived(const std::string& key) {
    for (size_t i = 0; i < pending_.size();) {
      if (pending_[i].key == key) {
        reply_value(pending_[i].fd, true, kv_[key]);
        pending_.erase(pending_.begin() + i);
      } else {
        ++i;
      }
    }
  }
  // Parse as many complete requests as `c.in` holds.
  void drain(Conn& c) {
    for (;;) {
      const std::string& b = c.in;
      if (b.size() < 1 + 4) return;
      const uint8_t op = (uint8_t)b[0];
      uint32_t klen;
      memcpy(&klen, b.data() + 1, 4);
      size_t need = 1 + 4 + klen;
      if (b.size() < need) return;
      std::string key = b.substr(5, klen);
      size_t consumed = need;
      if (op == OP_SET) {
        if (b.size() < need + 4) return;
        uint32_t vlen;
        memcpy(&vlen, b.data() + need, 4);
        if (b.size() < need + 4 + vlen) return;
        kv_[key] = b.substr(need + 4, vlen);
        consumed = need + 4 + vlen;
        reply_i64(c.fd, 0);
        key_arrived(key);
      } else if (op == OP_APPEND) {
        if (b.size() < need + 4) return;
        uint32_t vlen;
        memcpy(&vlen, b.data() + need, 4);
        if (b.size() < need + 4 + vlen) return;
        std::string& cur = kv_[key];
        cur.append(b, need + 4, vlen);
        consumed = need + 4 + vlen;
        reply_i64(c.fd, (int64_t)cur.size());
        key_arrived(key);
      } else if (op == OP_GET) {
        if (b.size() < need + 8) return;
        int64_t tmo;
        memcpy(&tmo, b.data() + need, 8);
        consumed = need + 8;
        auto it = kv_.find(key);
        if (it != kv_.end()) reply_value(c.fd, true, it->second);
        else if (tmo == 0) reply_value(c.fd, false, "");
        else pending_.push_back({c.fd, key, tmo < 0 ? Clock::time_point::max() : Clock::now() + std::chrono::milliseconds(tmo)});
      } else if (op == OP_ADD) {
        if (b.size() < need + 8) return;
        int64_t d;
        memcpy(&d, b.data() + need, 8);
        consumed = need + 8;
        int64_t cur = 0;
        auto it = kv_.find(key);
        if (it != kv_.end() && !it->second.empty()) cur = strtoll(it->second.c_str(), nullptr, 10);
        cur += d;
        kv_[key] = std::to_string(cur);
        reply_i64(c.fd, cur);
        key_arrived(key);
      } else if (op == OP_CHECK) {
        reply_i64(c.fd, kv_.count(key) ? 1 : 0);
      } else if (op == OP_DELETE) {
        reply_i64(c.fd, (int64_t)kv_.erase(key));
      } else if (op == OP_NUMKEYS) {
        reply_i64(c.fd, (int64_t)kv_.size());
      } else if (op == OP_CAS) {
        if (b.size() < need + 4) return;
        uint32_t elen;
        memcpy(&elen, b.data() + need, 4);
        if (b.size() < need + 4 + elen + 4) return;
        std::string expect = b.substr(need + 4, elen);
        uint32_t dlen;
        memcpy(&dlen, b.data() + need + 4 + elen, 4);
        if (b.size() < need + 4 + elen + 4 + dlen) return;
        std::string desired = b.substr(need + 4 + elen + 4, dlen);
        consumed = need + 4 + elen + 4 + dlen;
        auto it = kv_.find(key);
        if ((it == kv_.end() && expect.empty()) || (it != kv_.end() && it->second == expect)) {
          kv_[key] = desired;
          reply_value(c.fd, true, desired);
          key_arrived(key);
        } else {
          reply_value(c.fd, it != kv_.end(), it != kv_.end() ? it->second : "");
        }
      } else {
        c.in.clear();  // unknown op: drop the connection's buffer
        return;
      }
      c.in.erase(0, consumed);
    }
  }
  void loop() {
    while (running_) {
      std::vector<pollfd> pfs;
      pfs.push_back({lfd_, POLLIN, 0});
      pfs.push_back({wake_[0], POLLIN, 0});
      for (auto& c : conns_) pfs.push_back({c.fd, POLLIN, 0});
      int tmo = 200;
      int pr = ::poll(pfs.data(), pfs.size(), tmo);
      if (!running_) break;
      // expire timed-out GETs
      auto now = Clock::now();
      for (size_t i = 0; i < pending_.size();) {
        if (pending_[i].deadline <= now) {
          reply_value(pending_[i].fd, false, "");
          pending_.erase(pending_.begin() + i);
        } else {
          ++i;
        }
      }
      if (pr <= 0) continue;
      if (pfs[0].revents & POLLIN) {
        int fd = ::accept(lfd_, nullptr, nullptr);
        if (fd >= 0) {
          int one = 1;
          ::setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
          conns_.push_back({fd, ""});
        }
      }
      const size_t nconn = pfs.size() - 2;
      std::vector<int> dead;
      for (size_t i = 0; i < nconn && i < conns_.size(); ++i) {
        if (!(pfs[i + 2].revents & (POLLIN | POLLHUP | POLLERR))) continue;
        char buf[65536];
        ssize_t r = ::recv(conns_[i].fd, buf, sizeof(buf), 0);
        if (r <= 0) {
          dead.push_back(conns_[i].fd);
          continue;
        }
        conns_[i].in.append(buf, (size_t)r);
        drain(conns_[i]);
      }
      for (int fd : dead) {
        for (size_t i = 0; i < pending_.size();)
          if (pending_[i].fd == fd) pending_.erase(pending_.begin() + i);
          else ++i;
        for (size_t i = 0; i < conns_.size(); ++i)
          if (conns_[i].fd == fd) {
            ::close(fd);
            conns_.erase(conns_.begin() + i);
            break;
          }
      }
    }
  }

  int port_;
  int lfd_ = -1;
  int wake_[2] = {-1, -1};
  std::atomic<bool> running_{false};
  std::thread th_;
  std::vector<Conn> conns_;
  std::vector<Pending> pending_;
  std::map<std::string, std::string> kv_;
};

// ---------------------------------------------------------------- stores ----
class Store {
 public:
  virtual ~Store() {}
  virtual int set(const std::string& k, const std::string& v) = 0;
  virtual int get(const std::string& k, std::string* out, int64_t timeout_ms) = 0;  // 0 ok, 1 timeout, <0 error
  virtual int add(const std::string& k, int64_t d, int64_t* out) = 0;
  virtual int check(const std::string& k) = 0;
  virtual int del(const std::string& k) = 0;
  virtual int64_t num_keys() = 0;
  virtual int cas(const std::string& k, const std::string& expect, const std::string& desired, std::string* out) = 0;
  virtual int64_t append(const std::string& k, const std::string& v) = 0;  // new length of the value, <0 error
};

class TcpStore : public Store {
 public:
  TcpStore(const std::string& host, int port, bool is_server, int64_t timeout_ms)
      : host_(host), port_(port), timeout_ms_(timeout_ms) {
    if (is_server) {
      server_.reset(new TcpServer(port));
      if (!server_->start()) {
        server_.reset();
        ok_ = false;
        return;
      }
      port_ = server_->port();
    }
    ok_ = connect_client();
  }
  ~TcpStore() override {
    if (fd_ >= 0) ::close(fd_);
    server_.reset();
  }
  bool ok() const { return ok_; }
  int port() const { return port_; }

  int set(const std::string& k, const std::string& v) override {
    std::lock_guard<std::mutex> g(mu_);
    std::string m = head(OP_SET, k);
    put_str(m, v);
    int64_t r;
    return xfer_i64(m, &r);
  }
  int get(const std::string& k, std::string* out, int64_t timeout_ms) override {
    std::lock_guard<std::mutex> g(mu_);
    std::string m = head(OP_GET, k);
    put_i64(m, timeout_ms);
    if (!send_all(fd_, m.data(), m.size())) return -1;
    bool found;
    if (!recv_value(&found, out, timeout_ms < 0 ? -1 : (int)(timeout_ms + 5000))) return -1;
    return found ? 0 : 1;
  }
  int add(const std::string& k, int64_t d, int64_t* out) override {
    std::lock_guard<std::mutex> g(mu_);
    std::string m = head(OP_ADD, k);
    put_i64(m, d);
    return xfer_i64(m, out);
  }
  int check(const std::string& k) override {
    std::lock_guard<std::mutex> g(mu_);
    std::string m = head(OP_CHECK, k);
    int64_t r = 0;
    if (xfer_i64(m, &r) != 0) return -1;
    return (int)r;
  }
  int del(const std::string& k) override {
    std::lock_guard<std::mutex> g(mu_);
    std::string m = head(OP_DELETE, k);
    int64_t r = 0;
    if (xfer_i64(m, &r) != 0) return -1;
    return (int)r;
  }
  int64_t num_keys() override {
    std::lock_guard<std::mutex> g(mu_);
    std::string m = head(OP_NUMKEYS, "");
    int64_t r = 0;
    if (xfer_i64(m, &r) != 0) return -1;
    return r;
  }
  int64_t append(const std::string& k, const std::string& v) override {
    std::lock_guard<std::mutex> g(mu_);
    std::string m = head(OP_APPEND, k);
    put_str(m, v);
    int64_t r = -1;
    if (xfer_i64(m, &r) != 0) return -1;
    return r;
  }
  int cas(const std::string& k, const std::string& expect, const std::string& desired, std::string* out) override {
    std::lock_guard<std::mutex> g(mu_);
    std::string m = head(OP_CAS, k);
    put_str(m, expect);
    put_str(m, desired);
    if (!send_all(fd_, m.data(), m.size())) return -1;
    bool found;
    if (!recv_value(&found, out, (int)timeout_ms_)) return -1;
    return 0;
  }

 private:
  std::string head(Op op, const std::string& k) {
    std::string m;
    m.push_back((char)op);
    put_str(m, k);
    return m;
  }
  int xfer_i64(const std::string& m, int64_t* out) {
    if (!send_all(fd_, m.data(), m.size())) return -1;
    int64_t v;
    if (!recv_all(fd_, &v, 8, (int)timeout_ms_)) return -1;
    *out = v;
    return 0;
  }
  bool recv_value(bool* found, std::string* out, int tmo) {
    char f;
    if (!recv_all(fd_, &f, 1, tmo)) return false;
    uint32_t n;
    if (!recv_all(fd_, &n, 4, tmo)) return false;
    out->resize(n);
    if (n && !recv_all(fd_, &(*out)[0], n, tmo)) return false;
    *found = f != 0;
    return true;
  }
  bool connect_client() {
    addrinfo hints{}, *res = nullptr;
    hints.ai_family = AF_INET;
    hints.ai_socktype = SOCK_STREAM;
    const std::string ps = std::to_string(port_);
    auto deadline = Clock::now() + std::chrono::milliseconds(timeout_ms_ > 0 ? timeout_ms_ : 300000);
    while (Clock::now() < deadline) {
      if (::getaddrinfo(host_.c_str(), ps.c_str(), &hints, &res) == 0 && res) {
        int fd = ::socket(res->ai_family, res->ai_socktype, res->ai_protocol);
        if (fd >= 0) {
          if (::connect(fd, res->ai_addr, res->ai_addrlen) == 0) {
            int one = 1;
            ::setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
            fd_ = fd;
            ::freeaddrinfo(res);
            return true;
          }
          ::close(fd);
        }
        ::freeaddrinfo(res);
        res = nullptr;
      }
      std::this_thread::sleep_for(std::chrono::milliseconds(50));  // server may not be up yet
    }
    return false;
  }

  std::string host_;
  int port_;
  int64_t timeout_ms_;
  int fd_ = -1;
  bool ok_ = false;
  std::mutex mu_;
  std::unique_ptr<TcpServer> server_;
};

// FileStore: append-only log of (op, key, value) records; every operation takes an
// exclusive fcntl lock on the whole file, replays records appended since its last
// visit, then (for mutations) appends one record.
class FileStore : public Store {
 public:
  FileStore(const std::string& path, int64_t timeout_ms) : path_(path), timeout_ms_(timeout_ms) {
    fd_ = ::open(path.c_str(), O_RDWR | O_CREAT, 0644);
  }
  ~FileStore() override {
    if (fd_ >= 0) ::close(fd_);
  }
  bool ok() const { return fd_ >= 0; }

  int set(const std::string& k, const std::string& v) override {
    Lock l(fd_);
    refresh();
    append('S', k, v);
    return 0;
  }
  int get(const std::string& k, std::string* out, int64_t timeout_ms) override {
    auto deadline = timeout_ms < 0 ? Clock::time_point::max() : Clock::now() + std::chrono::milliseconds(timeout_ms);
    int backoff_us = 200;
    for (;;) {
      {
        Lock l(fd_);
        refresh();
        auto it = kv_.find(k);
        if (it != kv_.end()) {
          *out = it->second;
          return 0;
        }
      }
      if (Clock::now() >= deadline) return 1;
      ::usleep(backoff_us);
      if (backoff_us < 20000) backoff_us *= 2;
    }
  }
  int add(const std::string& k, int64_t d, int64_t* out) override {
    Lock l(fd_);
    refresh();
    int64_t cur = 0;
    auto it = kv_.find(k);
    if (it != kv_.end() && !it->second.empty()) cur = strtoll(it->second.c_str(), nullptr, 10);
    cur += d;
    append('S', k, std::to_string(cur));
    *out = cur;
    return 0;
  }
  int check(const std::string& k) override {
    Lock l(fd_);
    refresh();
    return kv_.count(k) ? 1 : 0;
  }
  int del(const std::string& k) override {
    Lock l(fd_);
    refresh();
    if (!kv_.count(k)) return 0;
    append('D', k, "");
    return 1;
  }
  int64_t num_keys() override {
    Lock l(fd_);
    refresh();
    return (int64_t)kv_.size();
  }
  int64_t append(const std::string& k, const std::string& v) override {
    Lock l(fd_);   // read-modify-write under the file lock: atomic across processes
    refresh();
    auto it = kv_.find(k);
    const std::string joined = (it != kv_.end() ? it->second : std::string()) + v;
    append('S', k, joined);
    return (int64_t)joined.size();
  }
  int cas(const std::string& k, const std::string& expect, const std::string& desired, std::string* out) override {
    Lock l(fd_);
    refresh();
    auto it = kv_.find(k);
    if ((it == kv_.end() && expect.empty()) || (it != kv_.end() && it->second == expect)) {
      append('S', k, desired);
      *out = desired;
    } else {
      *out = it != kv_.end() ? it->second : "";
    }
    return 0;
  }

 private:
  struct Lock {
    int fd;
    explicit Lock(int f) : fd(f) {
      struct flock fl{};
      fl.l_type = F_WRLCK;
      fl.l_whence = SEEK_SET;
      while (::fcntl(fd, F_SETLKW, &fl) != 0 && errno == EINTR) {}
    }
    ~Lock() {
      struct flock fl{};
      fl.l_type = F_UNLCK;
      fl.l_whence = SEEK_SET;
      ::fcntl(fd, F_SETLK, &fl);
    }
  };
  void refresh() {  // caller holds the lock
    struct stat st{};
    if (::fstat(fd_, &st) != 0) return;
    while (pos_ < st.st_size) {
      char hdr[9];
      if (::pread(fd_, hdr, 9, pos_) != 9) break;
      uint32_t kl, vl;
      memcpy(&kl, hdr + 1, 4);
      memcpy(&vl, hdr + 5, 4);
      if (pos_ + 9 + (off_t)kl + (off_t)vl > st.st_size) break;  // torn tail (writer died mid-append)
      std::string k(kl, '\0'), v(vl, '\0');
      if (kl && ::pread(fd_, &k[0], kl, pos_ + 9) != (ssize_t)kl) break;
      if (vl && ::pread(fd_, &v[0], vl, pos_ + 9 + kl) != (ssize_t)vl) break;
      if (hdr[0] == 'S') kv_[k] = v;
      else if (hdr[0] == 'D') kv_.erase(k);
      pos_ += 9 + kl + vl;
    }
  }
  void append(char op, const std::string& k, const std::string& v) {
    std::string rec;
    rec.push_back(op);
    put_u32(rec, (uint32_t)k.size());
    put_u32(rec, (uint32_t)v.size());
    rec.append(k);
    rec.append(v);
    (void)!::pwrite(fd_, rec.data(), rec.size(), pos_);
    ::fsync(fd_);
    pos_ += (off_t)rec.size();
    if (op == 'S') kv_[k] = v;
    else kv_.erase(k);
  }

  std::string path_;
  int64_t timeout_ms_;
  int fd_ = -1;
  off_t pos_ = 0;
  std::map<std::string, std::string> kv_;
};

}  // namespace

// ------------------------------------------------------------------ C ABI ----
PDA_EXPORT void* pda_store_tcp_create(const char* host, int port, int is_server, long long timeout_ms) {
  auto* s = new TcpStore(host, port, is_server != 0, timeout_ms);
  if (!s->ok()) {
    delete s;
    return nullptr;
  }
  return static_cast<Store*>(s);
}
PDA_EXPORT int pda_store_tcp_port(void* h) { return static_cast<TcpStore*>(static_cast<Store*>(h))->port(); }
PDA_EXPORT void* pda_store_file_create(const char* path, long long timeout_ms) {
  auto* s = new FileStore(path, timeout_ms);
  if (!s->ok()) {
    delete s;
    return nullptr;
  }
  return static_cast<Store*>(s);
}
PDA_EXPORT void pda_store_destroy(void* h) { delete static_cast<Store*>(h); }
PDA_EXPORT int pda_store_set(void* h, const char* key, const void* val, int len) {
  return static_cast<Store*>(h)->set(key, std::string(static_cast<const char*>(val), (size_t)len));
}
// Returns value length (>=0) and copies up to cap bytes; -1 timeout; -2 error. Call again with a
// larger buffer if the returned length exceeds cap.
PDA_EXPORT long long pda_store_get(void* h, const char* key, void* out, long long cap, long long timeout_ms) {
  std::string v;
  int r = static_cast<Store*>(h)->get(key, &v, timeout_ms);
  if (r == 1) return -1;
  if (r < 0) return -2;
  if (out && cap > 0) memcpy(out, v.data(), (size_t)std::min<long long>(cap, (long long)v.size()));
  return (long long)v.size();
}
PDA_EXPORT int pda_store_add(void* h, const char* key, long long delta, long long* out) {
  int64_t v = 0;
  int r = static_cast<Store*>(h)->add(key, delta, &v);
  *out = v;
  return r;
}
PDA_EXPORT int pda_store_check(void* h, const char* key) { return static_cast<Store*>(h)->check(key); }
PDA_EXPORT int pda_store_delete(void* h, const char* key) { return static_cast<Store*>(h)->del(key); }
PDA_EXPORT long long pda_store_num_keys(void* h) { return static_cast<Store*>(h)->num_keys(); }
PDA_EXPORT long long pda_store_append(void* h, const char* key, const void* val, int len) {
  return static_cast<Store*>(h)->append(key, std::string(static_cast<const char*>(val), (size_t)len));
}
PDA_EXPORT long long pda_store_compare_set(void* h, const char* key, const void* expect, int elen, const void* desired,
                                           int dlen, void* out, long long cap) {
  std::string v;
  int r = static_cast<Store*>(h)->cas(key, std::string(static_cast<const char*>(expect), (size_t)elen),
                                      std::string(static_cast<const char*>(desired), (size_t)dlen), &v);
  if (r < 0) return -2;
  if (out && cap > 0) memcpy(out, v.data(), (size_t)std::min<long long>(cap, (long long)v.size()));
  return (long long)v.size();
}
